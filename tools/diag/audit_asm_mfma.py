"""Audit of the kernels that issue MFMAs through inline asm with a TIED accumulator ("+v"): hipcc pads no wait states around
asm statements and knows nothing of the MFMA inside, so the properties the kernels rely on are checked on the compiler's
output (usage: python tools/diag/audit_asm_mfma.py [file.s ...]; without arguments conv3d.hip and wgrad.hip are compiled
to assembly for gfx950 first).  Per kernel that contains asm MFMAs:
  1. no scratch (a spilled accumulator would be re-loaded by compiler code next to an MFMA that wrote it);
  2. between the first and the last asm MFMA no other instruction reads or writes an accumulator register (inside the
     loops D is consumed only by the next MFMA of the same accumulator, whole, as C: no wait states needed);
  3. behind the last asm MFMA (text order) an instruction that touches an accumulator register is either part of the
     zero fill (moves of 0 / of a filled accumulator register: the block runs once, in front of the staging prologue,
     wherever the compiler places it) or sits behind the kernel's `s_nop 15` pair (MFMA D -> VALU read needs the wait
     states the compiler would have inserted itself)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'applying-slowfast-networks-to-video-object-segmentation_amd', 'csrc')


def compile_to_asm(src, out):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-S',
                    '--cuda-device-only', src, '-o', out], check=True, stderr=subprocess.DEVNULL)


def regs_of(line):
    r = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]', line):
        r.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', line):
        r.add(int(m.group(1)))
    return r


def audit(path):
    """-> list of (kernel, n_asm_mfma, problems)"""
    txt = open(path).read().split('\n')
    starts = [i for i, l in enumerate(txt) if re.match(r'^_Z\w+:', l)]
    out = []
    for k, st in enumerate(starts):
        name = txt[st].split(':')[0]
        end = starts[k + 1] if k + 1 < len(starts) else len(txt)
        body = txt[st:end]
        asm = [i for i, l in enumerate(body) if 'v_mfma' in l and i > 0 and ';;#ASMSTART' in body[i - 1]]
        if not asm:
            continue
        problems = []
        scratch = [l for l in body if re.search(r'\bscratch_(load|store)', l)]
        if scratch:
            problems.append('scratch accesses: %d' % len(scratch))
        accs = set()
        for i in asm:
            m = re.search(r'v\[(\d+):(\d+)\]', body[i])
            accs.update(range(int(m.group(1)), int(m.group(2)) + 1))
            ops = re.findall(r'v\[(\d+):(\d+)\]', body[i])
            if ops[0] != ops[3]:
                problems.append('accumulator not tied: ' + body[i].strip())
        lo, hi = asm[0], asm[-1]
        nop_at = next((i for i in range(hi, len(body)) if re.search(r'\bs_nop 15\b', body[i])), None)
        for i, l in enumerate(body):
            s = l.strip()
            if not s or s[0] in ';.' or s.endswith(':') or 'v_mfma' in s:
                continue
            if not (regs_of(s) & accs):
                continue
            # the zero fill: moves of the constant 0 / of an already filled accumulator register into accumulator registers
            # (the block may sit anywhere in the text; it runs once, in front of the staging prologue)
            fill = re.match(r'v_mov_b(32|64)_e32 (v\d+|v\[\d+:\d+\]), (0|v\d+|v\[\d+:\d+\])$', s) is not None and \
                regs_of(s) <= accs
            if lo < i < hi:
                problems.append('inside the MFMA region: ' + s)
            elif fill or i < lo:
                # in front of the first MFMA (text order): the prologue, where the registers still hold address arithmetic,
                # and the loop header, which cannot use them -- the accumulators are live through the whole loop
                continue
            elif nop_at is None or i < nop_at:
                problems.append('behind the last MFMA, not a fill, in front of the s_nop pair: ' + s)
        out.append((name, len(asm), problems))
    return out


def main(paths):
    tmp = None
    if not paths:
        tmp = tempfile.mkdtemp(prefix='sfvos_audit_')
        paths = []
        procs = []
        for f in ('conv3d.hip', 'wgrad.hip'):
            o = os.path.join(tmp, f.replace('.hip', '.s'))
            paths.append(o)
            hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
            procs.append(subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17',
                                           '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only',
                                           os.path.join(CSRC, f), '-o', o], stderr=subprocess.DEVNULL))
        for p in procs:
            if p.wait() != 0:
                raise RuntimeError('hipcc -S failed')
    bad = 0
    n = 0
    for p in paths:
        for name, cnt, problems in audit(p):
            n += 1
            print('%-100s %4d asm MFMAs  %s' % (name[:100], cnt, 'ok' if not problems else 'PROBLEMS'))
            for q in problems[:8]:
                print('      ' + q)
            bad += bool(problems)
    print('%d kernels with asm MFMAs, %d with problems' % (n, bad))
    return n, bad


if __name__ == '__main__':
    n, bad = main(sys.argv[1:])
    sys.exit(1 if bad or n == 0 else 0)
