import os, subprocess, sys
cases = {'f1': (22, ['5,1,0', '4,3,0', '3,5,0', '5,0,2', '4,2,2']),
         'f2': (12, ['3,0,0', '2,2,0', '1,4,0', '2,1,2', '0,6,0']),
         'df2': (22, ['5,1,0', '4,3,0', '3,5,0', '2,7,0', '0,11,0'])}
for name, (t, splits) in cases.items():
    for sp in splits:
        env = dict(os.environ, SFVOS_FS_SPLIT=sp)
        out = subprocess.run([sys.executable, 'tools/diag/mb_conv.py', name, '10'], env=env, capture_output=True, text=True).stdout
        print(name, sp, out.strip().splitlines()[-1] if out.strip() else 'FAILED', flush=True)
