"""Which HIP runtime(s) does a process hold when the library is loaded before / after torch?

ADVICE (round 2): torch.cuda's lazy initialisation reported "No HIP GPUs are available" when it came after library contexts
had been created. Hypothesis checked here: the torch wheel bundles its own libamdhip64.so / libhsa-runtime64.so (SONAME
libamdhip64.so.7, NEEDED by libc10_hip.so under the FILE name libamdhip64.so). Loaded first, that copy also satisfies
libdartray_rt.so's NEEDED libamdhip64.so.7 (SONAME match): one runtime in the process. Loaded second, glibc does not match
the file name against /opt/rocm's already loaded libamdhip64.so.7 and maps torch's copy too: two HIP + two HSA runtimes in
one process, and the second one finds no device.

    python tools/repro_torch_late_init.py            # runs the scenarios in child processes, prints a table
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dart-ray_amd", "csrc", "libdartray_rt.so")


def maps(pattern):
    out = set()
    with open("/proc/self/maps") as f:
        for line in f:
            p = line.split()[-1]
            if pattern in p:
                out.add(p)
    return sorted(out)


def hip_state(tag):
    h = ctypes.CDLL("libamdhip64.so.7")  # whatever copy the process already holds under that SONAME
    n = ctypes.c_int(-1)
    rc = h.hipGetDeviceCount(ctypes.byref(n))
    last = h.hipGetLastError()
    free, tot = ctypes.c_size_t(), ctypes.c_size_t()
    rc2 = h.hipMemGetInfo(ctypes.byref(free), ctypes.byref(tot))
    print(f"  [{tag}] hipGetDeviceCount rc={rc} n={n.value} hipGetLastError={last} hipMemGetInfo rc={rc2} free={free.value / 2**30:.1f} GiB", flush=True)


def contexts(n):
    L = ctypes.CDLL(LIB)
    L.drt_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    L.drt_destroy.argtypes = [ctypes.c_void_p]
    for _ in range(n):
        h = ctypes.c_void_p()
        rc = L.drt_create(ctypes.byref(h), 0)
        assert rc == 0, rc
        L.drt_destroy(h)


def scenario(name):
    if name == "lib_then_torch":
        contexts(12)
        hip_state("after 12 contexts")
        import torch
        try:
            torch.cuda.init()
            print("  torch.cuda.init() ok, devices:", torch.cuda.device_count())
        except Exception as e:
            print("  torch.cuda.init() FAILED:", str(e).splitlines()[0])
    elif name == "one_context_then_torch":
        contexts(1)
        import torch
        try:
            torch.cuda.init()
            print("  torch.cuda.init() ok, devices:", torch.cuda.device_count())
        except Exception as e:
            print("  torch.cuda.init() FAILED:", str(e).splitlines()[0])
    elif name == "import_torch_then_lib_then_init":
        import torch  # maps torch's runtime, initialises nothing
        contexts(12)
        hip_state("after 12 contexts")
        try:
            torch.cuda.init()
            print("  torch.cuda.init() ok, devices:", torch.cuda.device_count())
        except Exception as e:
            print("  torch.cuda.init() FAILED:", str(e).splitlines()[0])
    elif name == "host_py_then_torch":
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge
        pkg = ge._load_pkg()
        pkg.host.lib()  # the product's loader
        contexts(12)
        import torch
        try:
            torch.cuda.init()
            print("  torch.cuda.init() ok, devices:", torch.cuda.device_count())
        except Exception as e:
            print("  torch.cuda.init() FAILED:", str(e).splitlines()[0])
    print("  libamdhip64 copies mapped:", maps("libamdhip64"))
    print("  libhsa-runtime64 copies mapped:", maps("libhsa-runtime64"))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        scenario(sys.argv[1])
    else:
        for s in ("lib_then_torch", "one_context_then_torch", "import_torch_then_lib_then_init", "host_py_then_torch"):
            print(s, flush=True)
            subprocess.call([sys.executable, os.path.abspath(__file__), s])
