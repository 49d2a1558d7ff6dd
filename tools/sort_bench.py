#!/usr/bin/env python
"""Microbenchmark of the 64-bit-key radix sort (one digit pass = 24*n bytes)."""
import ctypes as ct
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run_one(n, reps):
    import torch
    from boxtree_amd import HIPArrayContext, _lib
    actx = HIPArrayContext(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    keys0 = torch.randint(0, 2**62, (n,), generator=g, dtype=torch.int64, device="cuda")
    vals0 = torch.arange(n, dtype=torch.int32, device="cuda")
    best = 1e9
    for _ in range(reps):
        k, v = keys0.clone(), vals0.clone()
        ko, vo = torch.empty_like(k), torch.empty_like(v)
        torch.cuda.synchronize()
        _lib.check(actx.lib.bt_radix_sort_u64_u32(
            actx.handle, ct.c_void_p(k.data_ptr()), ct.c_void_p(v.data_ptr()),
            ct.c_void_p(ko.data_ptr()), ct.c_void_p(vo.data_ptr()), n, 0, 64))
        st = _lib.SortStats()
        actx.lib.bt_get_sort_stats(actx.handle, st)
        best = min(best, st.pass_ms_avg)
    ok = bool((ko[1:] >= ko[:-1]).all())
    print(f"cfg={os.environ.get('BT_SORT_CFG', '0')} n={n} pass_ms={best:.4f} "
          f"GB/s={24 * n / best / 1e6:.0f} frac={24 * n / best / 1e6 / 8000:.3f} "
          f"hist_ms={st.hist_ms:.3f} sorted={ok}", flush=True)


def run_keys(n, reps, path_bits=36):
    """The keys-only sort of a packed-key tree build: path bits over ceil(log2 n) id bits."""
    import torch
    from boxtree_amd import HIPArrayContext, _lib
    actx = HIPArrayContext(0)
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    idbits = max(1, (n - 1).bit_length())
    keys0 = (torch.randint(0, 2**path_bits, (n,), generator=g, dtype=torch.int64, device="cuda")
             << idbits) | torch.arange(n, dtype=torch.int64, device="cuda")
    best = 1e9
    for _ in range(reps):
        k = keys0.clone()
        ko = torch.empty_like(k)
        torch.cuda.synchronize()
        _lib.check(actx.lib.bt_radix_sort_u64_keys(
            actx.handle, ct.c_void_p(k.data_ptr()), ct.c_void_p(ko.data_ptr()), n, idbits,
            idbits + path_bits))
        st = _lib.SortStats()
        actx.lib.bt_get_sort_stats(actx.handle, st)
        best = min(best, st.pass_ms_avg)
    # (path, id) ascending as 64-bit words == stable by path
    ok = bool((ko[1:] > ko[:-1]).all())
    bpe = st.bytes_per_element_per_pass
    print(f"keys-only kcfg={os.environ.get('BT_SORT_KCFG', '0')} n={n} path_bits={path_bits} "
          f"digit_bits={st.digit_bits} passes={st.passes} pass_ms={best:.4f} "
          f"GB/s={bpe * n / best / 1e6:.0f} frac={bpe * n / best / 1e6 / 8000:.3f} "
          f"hist_ms={st.hist_ms:.3f} total_ms={st.total_ms:.3f} sorted={ok}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        run_one(int(sys.argv[2]), int(sys.argv[3]))
    elif len(sys.argv) > 1 and sys.argv[1] == "keys":
        run_keys(int(sys.argv[2]), int(sys.argv[3]), *[int(a) for a in sys.argv[4:5]])
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 10**8
        cfgs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "1", "2", "3", "4", "5"]
        for c in cfgs:
            env = dict(os.environ, BT_SORT_CFG=c)
            subprocess.call([sys.executable, __file__, "one", str(n), "3"], env=env)
