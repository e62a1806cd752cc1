"""Host-memory settings of the command-line entry points (train / train_sketch / eval).

The hosts these run on are virtual machines on which a first touch of a page is expensive (a fresh 4 MB host tensor: 6 - 20 ms, measured
with tools/probe/eval_profile.sh; the arithmetic that fills it: 2 ms).  glibc hands every block above 128 KiB to mmap and returns it to the
kernel on free, so every collated batch, every float64 intermediate of --add_noise, every (B,K,S) draw buffer is first-touched again.
retain_large_blocks() raises the mmap threshold to its maximum (32 MiB) and the trim threshold to 2 GiB - 1 (mallopt takes an int): such
blocks then come from the heap and stay mapped after free (the resident size of a long run only grows up to its largest working set).
P2C_HOSTMEM=0 leaves the allocator and torch's thread pool alone.  Process-wide, so only the CLIs call it - a library has no business changing its host's allocator."""
import ctypes
import os

_M_TRIM_THRESHOLD, _M_TOP_PAD, _M_MMAP_THRESHOLD = -1, -2, -3


def retain_large_blocks():
    """-> True when glibc took both settings (False on another libc: nothing changed)."""
    try:
        libc = ctypes.CDLL(None)
        mallopt = libc.mallopt
    except (OSError, AttributeError):
        return False
    mallopt.argtypes, mallopt.restype = [ctypes.c_int, ctypes.c_int], ctypes.c_int
    ok = mallopt(_M_MMAP_THRESHOLD, 32 << 20) == 1
    ok = mallopt(_M_TRIM_THRESHOLD, 2**31 - 1) == 1 and ok
    mallopt(_M_TOP_PAD, 64 << 20)
    return ok


def cpu_quota():
    """The CPUs this process may actually use: the cgroup's quota (cpu.max, v2; cfs_quota_us / cfs_period_us, v1) and its affinity mask,
    whichever is smaller - os.cpu_count() reports the machine's (256 on the hosts this runs on, under a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def fit_threads_to_quota(reserve=2):
    """torch's intra-op pool sized to this rank's share of the CPU quota (quota / LOCAL_WORLD_SIZE) minus `reserve` (the launching thread
    and the loader thread): a pool of one thread per visible core under a 16-CPU quota spends its slices being throttled - every CPU-side
    torch op of the evaluation loop (collate, casts, draws) then costs milliseconds (tools/probe/eval_profile.sh: 83 -> 5 ms a batch).
    -> the pool size set."""
    import torch
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    n = max(1, min(torch.get_num_threads(), cpu_quota() // ranks - reserve))
    torch.set_num_threads(n)
    return n


def setup_cli():
    """Called by the CLIs (train / train_sketch / eval / bench.py) and tests/conftest.py - never by the library.  P2C_HOSTMEM=0: no-op."""
    if os.environ.get("P2C_HOSTMEM", "1") == "0":
        import torch
        return torch.get_num_threads()
    retain_large_blocks()
    return fit_threads_to_quota()
