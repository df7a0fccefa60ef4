"""Run scripts/probes/lt_tune_probe.hip's tuner INSIDE a process that imported torch, so that the hipBLASLt that answers is the
build the product binds to (torch/lib/libhipblaslt.so - loaded first, same soname as /opt/rocm's).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DLT_TUNE_AS_LIBRARY scripts/probes/lt_tune_probe.hip \
          -o scripts/lablib/liblt_tune.so -L/opt/rocm/lib -lhipblaslt
    python scripts/debug/lt_tune.py M N K [out 0|1] [max seconds] [M2 ...]
Record: profiles/r06_lt_tune.txt"""
import ctypes, os, sys

os.environ.setdefault("TENSILE_STREAMK_DATA_PARALLEL", "1")
import torch

torch.cuda.init()
torch.zeros(1, device="cuda")                      # the HIP runtime and torch's libhipblaslt are in the process now
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(here, "lablib", "liblt_tune.so"))
args = [b"lt_tune"] + [a.encode() for a in sys.argv[1:]]
argv = (ctypes.c_char_p * len(args))(*args)
sys.stdout.flush()
sys.exit(lib.lt_tune_main(len(args), argv))
