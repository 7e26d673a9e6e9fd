#!/usr/bin/env python3
"""Experiment (diagnostic build: JCDF_LIB_PATH=tools/_build/libjcdf_hip_diag.so): does a keep-alive on CUs the eigensolve does
NOT use hold the shader clock for the next Fock build without slowing the eigensolve?  The eigensolve runs on a stream masked
to a subset of the CUs (hipExtStreamCreateWithCUMask), FMA-chain keep-alive waves on a stream masked to the others.
usage: cumask_test.py [eig_cus=96]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder
from juliachem_jl_amd.eigh import DeviceEigh
N, Q, o = synthetic.CONFIGS["C20H42"]
eig_cus = int(sys.argv[1]) if len(sys.argv) > 1 else 96
rng = np.random.default_rng(1); dev = torch.device("cuda", 0)
fb = DeviceFockBuilder(N, Q, o, [1] * Q, device=0)
g = torch.Generator(device=dev); g.manual_seed(7)
for c0 in range(0, N * N, 16384):
    c1 = min(N * N, c0 + 16384)
    blk = torch.randn((c1 - c0, len(fb.rows)), dtype=torch.float64, device=dev, generator=g) * 0.05
    fb.h.set_B_columns_device(c0, c1, blk.data_ptr())
fb.set_core_hamiltonian(np.eye(N))
C, _ = np.linalg.qr(rng.standard_normal((N, N)))
Ct = torch.as_tensor(np.ascontiguousarray(C[:, :o].T), device=dev)
eig = DeviceEigh(N, dev)
S = torch.randn((N, N), dtype=torch.float64, device=dev); S = S + S.T
lib = jc._lib.load()
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
sink = torch.zeros(8, dtype=torch.float64, device=dev)
stop = torch.zeros(4, dtype=torch.int32, device=dev)


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


# two ways of choosing the eigensolve's CUs: the first `eig_cus` bits, or every k-th bit
sets = {"low": list(range(eig_cus)), "strided": [b for b in range(256) if b % 8 < eig_cus // 32]}
main = torch.cuda.current_stream(dev)


def run(label, pre, reps=12):
    for _ in range(3):
        pre(); fb.build(Ct)
    torch.cuda.synchronize()
    fb.h.kernel_stats_total(reset=True)
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); pre(); e1.record(); evs.append((e0, e1))
        fb.build(Ct)
    torch.cuda.synchronize()
    recs, n, fock = fb.h.kernel_stats_total(reset=True)
    k = {r["name"]: r["seconds"] / max(n, 1) * 1e3 for r in recs}
    print("%-44s W %.3f K %.3f fock %.3f | pre %.3f ms" % (label, k["k_exchange_W"], k["k_exchange_K"], fock / max(n, 1) * 1e3,
                                                            sum(a.elapsed_time(b) for a, b in evs) / reps), flush=True)


run("back to back", lambda: None)
run("eigensolve (all CUs, as shipped)", lambda: eig(S))
for name, bits in sets.items():
    se = masked_stream(bits)
    sk = masked_stream([b for b in range(256) if b not in set(bits)])

    def eig_masked(keep_mode):
        ev = torch.cuda.Event(); ev.record(main)
        se.wait_event(ev)
        if keep_mode is not None:
            stop.zero_()
            sk.wait_event(ev)
            wg, thr = keep_mode
            rc = lib.jcdf_keepalive_device(ctypes.c_void_p(sk.cuda_stream), wg, thr, 20000.0, 1, 0, ctypes.c_void_p(stop.data_ptr()),
                                           ctypes.c_void_p(sink.data_ptr()))
            assert rc == 0
        with torch.cuda.stream(se):
            eig(S)
            if keep_mode is not None:
                stop.fill_(1)
        main.wait_stream(se)
        if keep_mode is not None:
            main.wait_stream(sk)
    run("eigensolve on %d CUs (%s), nothing beside" % (len(bits), name), lambda: eig_masked(None))
    for km in ((160, 256), (320, 256), (640, 256)):
        run("  + FMA chains %d x %d on the other CUs" % km, lambda km=km: eig_masked(km))
