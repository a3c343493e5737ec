"""Probe (GPU box): is the persistent GEMM power-bound?  The SAME kernel (gemm_pp PP_GELU, M = 65536, N = 3072, K = 768: the FFN-1
launch) on random operands, on operands of small dynamic range (few toggling mantissa bits) and on all-zero operands: instruction
stream, LDS / DMA traffic and synchronisation are identical, only the switching activity of the data differs.
    python tools/gemm_power_probe.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memvul_amd.binding import Engine  # noqa: E402

M, N, K = 65536, 3072, 768
rng = np.random.default_rng(2021)
eng = Engine(0, vocab_size=1024, layers=1, max_tokens=M, max_batch=256, max_anchors=8)
bias = np.zeros(N, np.float32)
cases = {
    "random N(0,1) x N(0,0.03)": (rng.standard_normal((M, K), np.float32), rng.standard_normal((N, K), np.float32) * 0.03),
    "powers of two (+-1, +-2^-5)": (np.sign(rng.standard_normal((M, K))).astype(np.float32), np.sign(rng.standard_normal((N, K))).astype(np.float32) / 32),
    "constant 1.0 x constant 2^-5": (np.ones((M, K), np.float32), np.full((N, K), 1 / 32, np.float32)),
    "zeros": (np.zeros((M, K), np.float32), np.zeros((N, K), np.float32)),
}
flops = 2.0 * M * N * K
for rep in range(2):
    for name, (A, W) in cases.items():
        for x8 in (False, True):
            _, _, ms = eng.test_gemm_pp(A, W, bias, x8=x8, iters=20)
            print(f"{name:32s} {'precise' if x8 else 'f16    '}  {ms * 1e3:8.1f} us  {flops / ms / 1e9:8.1f} TF (algorithmic)", flush=True)
