#!/usr/bin/env python3
"""Diagnostic: board power and shader clock (rocm-smi, sampled from a side process) while one GEMM runs back to back for a few
seconds - ours (phase-staggered tap-GEMM) and the library yardstick (torch.addmm = hipBLASLt; never on the product path) at the
encoder FC2 / QKV shapes, random and zero operands.  Answers whether both sit at the same power cap (then the difference is
energy per FLOP) or at different clocks for another reason.
usage: python tools/gemm_power.py [clips=640] [seconds=3]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.ops import F_RES_POST

CLIPS = int(sys.argv[1]) if len(sys.argv) > 1 else 640
SECS = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
M = CLIPS * 100


def sample(stop, rows):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "-P", "-c", "--json"], capture_output=True, text=True, timeout=10).stdout
            pw = re.search(r'"[^"]*[Pp]ower[^"]*\(W\)"\s*:\s*"([\d.]+)"', out)
            ck = re.search(r'"sclk clock speed:"\s*:\s*"\((\d+)Mhz\)"', out)
            rows.append((float(pw.group(1)) if pw else None, int(ck.group(1)) if ck else None, out if not pw else None))
        except Exception as e:   # the tool may be missing or unreadable for this user: say so once
            rows.append((None, None, repr(e)))
            return
        time.sleep(0.2)


def measure(name, run, flops):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, rows))
    th.start()
    n = 0
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECS:
        for _ in range(20):
            run()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    pws = [r[0] for r in rows[1:] if r[0] is not None]
    cks = [r[1] for r in rows[1:] if r[1] is not None]
    pw = f"{sum(pws) / len(pws):6.0f} W (max {max(pws):.0f})" if pws else "power n/a"
    ck = f"{sum(cks) / len(cks):5.0f} MHz" if cks else "sclk n/a"
    print(f"{name:34s} {us:8.1f} us {flops / us / 1e6:7.0f} TF   {pw}   {ck}   [{len(rows)} samples]", flush=True)
    if not pws and rows:
        print("   rocm-smi said:", str(rows[0][2])[:300])


for shape, N, K in (("fc2", 1024, 4096), ("qkv", 3072, 1024)):
    for data in ("random", "zeros"):
        mk = torch.randn if data == "random" else (lambda *s, device: torch.zeros(*s, device=device))
        a = mk(M, K, device="cuda").half()
        w = (mk(N, K, device="cuda") / K ** 0.5).half()
        b = torch.zeros(N, device="cuda")
        b16 = b.half()
        fl = 2.0 * M * N * K
        if shape == "fc2":
            x = torch.zeros(M, N, device="cuda")
            ours = lambda: ops.tapgemm(a, w, x, M=M, N=N, Cin=K, bias=b, R=x, ldr=N, flags=F_RES_POST, dtype=ops.F16)
        else:
            c = torch.empty(M, N, device="cuda", dtype=torch.float16)
            ours = lambda: ops.tapgemm(a, w, c, M=M, N=N, Cin=K, bias=b, dtype=ops.F16)
        wt = w.t()
        measure(f"{shape} {data:6s} ours", ours, fl)
        measure(f"{shape} {data:6s} lib (hipBLASLt)", lambda: torch.addmm(b16, a, wt), fl)
