"""CPU oracle for the SlowFastLayers hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline -- never as a compute
path of ``sfvos_amd``.  The product fails loudly when ``libsfvos.so`` (the HIP
library) is missing; it never falls back to this code.

Pinning: ``oracle/slowfast_ref.py`` is a restatement, in torch-CPU functional
ops, of ``/root/reference/code/helpers/model.py:30-165``.  It is pinned against
the reference's own class (imported unmodified in the build container by
``oracle/make_golden.py``) through the fixtures under ``tests/golden/``.
"""
