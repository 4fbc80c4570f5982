"""Runner of tests/test_bench_multirank.py: bench.py's multi-rank control flow on CPU.  TEST INFRASTRUCTURE ONLY.
Injects the emulated library (tests/emu: the product's api.cc + kernels over a CPU model of the wavefront) where the
product binds libspmx.so, sets bench.py's dry-run seam (CPU tensors, gloo) and calls bench.main() with the arguments the
driver passes.  SPMX_DRYRUN_HANG=<algo> makes that gather algorithm block for ever (the watchdog's case)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPMX_BENCH_DRYRUN"] = "1"
os.environ["LOCAL_RANK"] = "0"                   # (the emulated runtime has one device; RANK / WORLD_SIZE stay)
os.environ.setdefault("SPMX_EMU_CUS", "2")
os.environ.setdefault("SPMX_UNI_WAVE_MAX", "0")

from sentencepiece_amd import _capi, sharding   # noqa: E402
from tests import emulib                         # noqa: E402

_capi.lib = emulib.lib                           # the processor's default binding -> the emulated library

hang = os.environ.get("SPMX_DRYRUN_HANG")
if hang:
    real_call = sharding.IdGatherer.__call__

    def call(self, ids, total, id_offsets=None, **kw):
        if self.algo == hang:
            time.sleep(3600)
        return real_call(self, ids, total, id_offsets, **kw)
    sharding.IdGatherer.__call__ = call

import bench  # noqa: E402

bench.main()
