"""CPU emulation of attention_v2.h's index algebra (memvul_amd/csrc/attention_v2.h): LDS-DMA placement with the
source-side swizzle, fragment read addresses, the pi(i) key order of the swapped QK^T, the P^T / V^T k-slot pairing,
and the O image used for whole-row stores.  The emulator restates the kernel's address formulas and the
v_mfma_f32_32x32x16_f16 operand / result layouts (cdna_hip_programming.md §3) in numpy and checks the result against
a plain softmax(QK^T)V — it guards the layout logic that cannot be run without a GPU (synchronisation is not
modelled).  The GPU parity tests check the real kernel."""
import numpy as np
import pytest


def mfma_32x32x16(a_frag, b_frag, c):
    """a_frag[lane][j], b_frag[lane][j] (j = 0..7), c[lane][r] (r = 0..15) -> c + A B with
    A[i][k]: lane = i + 32 * (k // 8), element k % 8;  B[k][n]: lane = n + 32 * (k // 8), element k % 8;
    C[i][n]: lane = n + 32 * hi, register r with i = (r & 3) + 8 * (r >> 2) + 4 * hi."""
    A = np.zeros((32, 16), np.float32)
    B = np.zeros((16, 32), np.float32)
    for lane in range(64):
        for j in range(8):
            A[lane & 31, 8 * (lane >> 5) + j] = a_frag[lane][j]
            B[8 * (lane >> 5) + j, lane & 31] = b_frag[lane][j]
    C = A @ B
    out = c.copy()
    for lane in range(64):
        hi = lane >> 5
        for r in range(16):
            out[lane][r] += C[(r & 3) + 8 * (r >> 2) + 4 * hi, lane & 31]
    return out


def emulate_item(K, VT, Q, length, NKB):
    """K [S][64], VT [64][S], Q [S][64] (fp16 values as float32) -> ctx rows [S][64] as the kernel would store them."""
    S = 64 * NKB
    NT = 2 * NKB
    VOFF = NKB * 8192
    lds = np.zeros(NKB * 16384 // 2, np.float32)  # one ring slot, indexed in halfs (2 bytes)

    def lds_write16(byte_addr, vals8):
        assert byte_addr % 16 == 0
        lds[byte_addr // 2: byte_addr // 2 + 8] = vals8

    def lds_read16(byte_addr):
        assert byte_addr % 16 == 0
        return lds[byte_addr // 2: byte_addr // 2 + 8].copy()

    Kb = K.reshape(-1)      # byte offset o -> half index o // 2
    Vb = VT.reshape(-1)
    nwaves = 2 * NKB
    # ---- LDS-DMA: wave-uniform LDS base + lane * 16, per-lane global source
    for wave in range(nwaves):
        for lane in range(64):
            rl = lane >> 3
            srcK = [rl * 128 + ((lane & 7) ^ (((rl >> 1) + 4 * x) & 7)) * 16 for x in range(2)]
            srcV = [rl * (2 * S) + ((lane & 7) ^ (((rl >> 1) + 4 * x) & 7)) * 16 for x in range(2)]
            for x in range(4):
                p = 4 * wave + x
                src = p * 1024 + srcK[x & 1]
                lds_write16(p * 1024 + lane * 16, Kb[src // 2: src // 2 + 8])
                src = (8 * (p & 7)) * (2 * S) + (p >> 3) * 128 + srcV[x & 1]
                lds_write16(VOFF + p * 1024 + lane * 16, Vb[src // 2: src // 2 + 8])
    out = np.zeros((S, 64), np.float32)
    o_img = np.zeros(S * 128 // 2, np.float32)
    for wave in range(nwaves):
        lanes = range(64)
        hi = [l >> 5 for l in lanes]
        ql = [l & 31 for l in lanes]
        pq = [(q & 0x13) | ((q & 4) << 1) | ((q & 8) >> 1) for q in ql]
        koff = [[pq[l] * 128 + (((2 * x + hi[l]) ^ ((pq[l] >> 1) & 7)) << 4) for x in range(4)] for l in lanes]
        voff = [[ql[l] * 128 + (((2 * x + hi[l]) ^ ((ql[l] >> 1) & 7)) << 4) for x in range(4)] for l in lanes]
        qf = [[Q[32 * wave + ql[l], 16 * kk + 8 * hi[l]: 16 * kk + 8 * hi[l] + 8] for l in lanes] for kk in range(4)]
        st = []
        for t in range(NT):
            c = np.zeros((64, 16), np.float32)
            for kk in range(4):
                kf = [lds_read16(t * 4096 + koff[l][kk]) for l in lanes]
                c = mfma_32x32x16(kf, qf[kk], c)
            st.append(c)
        st = np.stack(st)  # [t][lane][r]
        if length < S:
            for l in lanes:
                thr = length - 8 * hi[l]
                for t in range(NT):
                    for r in range(16):
                        if 32 * t + 16 * (r >> 3) + 4 * ((r >> 2) & 1) + (r & 3) >= thr:
                            st[t, l, r] += -10000.0
        mx = st.max(axis=(0, 2))
        mx = np.maximum(mx, mx[[l ^ 32 for l in lanes]])
        p = np.exp(st - mx[None, :, None]).astype(np.float32)
        psum = p.sum(axis=(0, 2))
        inv = 1.0 / (psum + psum[[l ^ 32 for l in lanes]])
        p16 = p.astype(np.float16).astype(np.float32)
        o = [np.zeros((64, 16), np.float32) for _ in range(2)]
        for t in range(NT):
            for u in range(2):
                pf = [p16[t, l, 8 * u: 8 * u + 8] for l in lanes]
                for dt in range(2):
                    vf = [lds_read16(VOFF + (t >> 1) * 8192 + dt * 4096 + voff[l][2 * (t & 1) + u]) for l in lanes]
                    o[dt] = mfma_32x32x16(vf, pf, o[dt])
        # ---- O image: [dt][rg] 8-byte groups, row 32 wave + ql, slot (4 dt + rg) ^ (ql & 7), half hi
        for l in lanes:
            o_wr = (32 * wave + ql[l]) * 128 + 8 * hi[l]
            for dt in range(2):
                for rg in range(4):
                    addr = o_wr ^ (((4 * dt + rg) ^ (ql[l] & 7)) << 4)
                    vals = (o[dt][l, 4 * rg: 4 * rg + 4] * inv[l]).astype(np.float16).astype(np.float32)
                    o_img[addr // 2: addr // 2 + 4] = vals
        for l in lanes:
            o_rd = (32 * wave + (l >> 3)) * 128 + (((l & 7) ^ (l >> 3)) << 4)
            for it in range(4):
                row = 32 * wave + (l >> 3) + 8 * it
                a = o_rd + it * 1024
                out[row, 8 * (l & 7): 8 * (l & 7) + 8] = o_img[a // 2: a // 2 + 8]
    return out


@pytest.mark.parametrize("NKB,length", [(1, 64), (2, 128), (4, 256), (4, 201), (3, 7)])
def test_attention_v2_index_algebra(NKB, length):
    S = 64 * NKB
    rng = np.random.default_rng(100 * NKB + length)
    K = rng.standard_normal((S, 64)).astype(np.float16).astype(np.float32)
    V = rng.standard_normal((S, 64)).astype(np.float16).astype(np.float32)
    Q = (rng.standard_normal((S, 64)) * 0.5).astype(np.float16).astype(np.float32)
    got = emulate_item(K, np.ascontiguousarray(V.T), Q, length, NKB)
    s = Q @ K.T
    s[:, length:] += -10000.0
    p = np.exp(s - s.max(axis=1, keepdims=True))
    ref = (p / p.sum(axis=1, keepdims=True)) @ V
    assert np.abs(got - ref).max() < 6e-3  # fp16 rounding of P and of the stored context


@pytest.mark.parametrize("nch", [2, 3, 4])
@pytest.mark.parametrize("heads_total", [12, 36, 60, 128 * 12, 7 * 12])
def test_chunked_work_list_is_a_permutation_that_co_locates_query_blocks(nch, heads_total):
    """attention_v2_kernel<., NCH > 1>: list position -> (head, query block).  Every unit exactly once; inside full
    groups of 8 NCH positions the NCH query blocks of a head are 8 positions apart (same XCD under round-robin dispatch)."""
    nunits = heads_total * nch
    nfull = nunits - nunits % (8 * nch)

    def unit(u):
        if u >= nfull:
            return u // nch, u % nch
        g, r = divmod(u, 8 * nch)
        return 8 * g + (r & 7), r >> 3

    seen = {unit(u) for u in range(nunits)}
    assert len(seen) == nunits and seen == {(bh, qb) for bh in range(heads_total) for qb in range(nch)}
    for u in range(0, nfull - 8):
        bh, qb = unit(u)
        if qb + 1 < nch:
            assert unit(u + 8) == (bh, qb + 1)
