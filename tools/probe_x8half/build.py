#!/usr/bin/env python3
"""Round-4 TIMING probes of the precise mode's correction sweep (numerically meaningless; nothing here ships).
Builds libraries from a scratch copy of memvul_amd/csrc with gemm_pp.h patched:
  half   : the fp8 sweep stages HALF the bytes (one LDS-DMA piece per wave and half-tile instead of two, counted wait halved),
           reads half the fragments (chunk q4 only, used twice) and pays the in-register "hi8 from the fp16 fragment" VALU
           (8 v_pk_add_u16 + 4 v_perm_b32 per 16 values) — what an fp8 sweep costs when only the lo8 planes travel.
  half6  : the same with the fp6 issue rate on the sweep's MFMAs (cbsz:2 blgp:2) — the floor of an MX-fp6 correction.
  nocorr : the fp8 sweep removed (nseg = 1 in an X8 build): what the precise mode's epilogues / planes cost without the sweep.
usage: python tools/probe_x8half/build.py  -> tools/probe_x8half/lib_{half,half6,nocorr}.so
"""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))

def patch(src, mode):
    s = src
    if mode == "halfbar":  # every other barrier (and counted wait) of the fp8 sweep removed: is the sweep bound by its per-interval synchronisation?
        old = '''    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };'''
        new = '''    if constexpr (!(X8 && decltype(f8c)::value && (s & 1) == 0)) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    }
  };'''
        assert old in s
        return s.replace(old, new)
    if mode == "nobar8":  # NO barrier / wait in the fp8 sweep at all (timing only)
        old = '''    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };'''
        new = '''    if constexpr (!(X8 && decltype(f8c)::value)) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    }
  };'''
        assert old in s
        return s.replace(old, new)
    if mode == "nocorr":
        s = s.replace("constexpr int nseg = X8 ? 2 : 1;", "constexpr int nseg = 1;")
        s = s.replace("two_ktiles(std::false_type{}, !X8 && kt + 2 >= nk0);", "two_ktiles(std::false_type{}, kt + 2 >= nk0);")
        assert s != src
        return s
    # (1) one DMA piece per wave and half-tile in the fp8 segment
    s = s.replace("      glds16((const half_t*)(src + offA[0]), dst);\n      glds16((const half_t*)(src + offA[1]), dst + 1024);",
                  "      glds16((const half_t*)(src + offA[0]), dst);\n      if (!(X8 && i_seg)) glds16((const half_t*)(src + offA[1]), dst + 1024);")
    s = s.replace("      glds16((const half_t*)(src + offB[0]), dst);\n      glds16((const half_t*)(src + offB[1]), dst + 1024);",
                  "      glds16((const half_t*)(src + offB[0]), dst);\n      if (!(X8 && i_seg)) glds16((const half_t*)(src + offB[1]), dst + 1024);")
    # (2) counted wait per format
    s = s.replace('asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");\n    __builtin_amdgcn_sched_barrier(0);\n    __builtin_amdgcn_s_barrier();\n    __builtin_amdgcn_sched_barrier(0);\n  };',
                  'if constexpr (X8 && decltype(f8c)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN / 2) : "memory");\n    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");\n    __builtin_amdgcn_sched_barrier(0);\n    __builtin_amdgcn_s_barrier();\n    __builtin_amdgcn_sched_barrier(0);\n  };')
    # (3) half the fragment reads + the conversion VALU in the fp8 segment: read_a / read_b get the format flag through a member
    s = s.replace("  auto read_a = [&](int par, int asub) {", "  bool rd_f8 = false;\n  auto cvt_cost = [&](intx4 v) -> intx4 {\n    uint32_t a = v[0], b = v[1], c = v[2], d = v[3];\n    asm volatile(\"v_pk_add_u16 %0, %0, %4\\n\\tv_pk_add_u16 %1, %1, %4\\n\\tv_pk_add_u16 %2, %2, %4\\n\\tv_pk_add_u16 %3, %3, %4\\n\\t\"\n                 \"v_pk_add_u16 %0, %0, %4\\n\\tv_pk_add_u16 %1, %1, %4\\n\\tv_pk_add_u16 %2, %2, %4\\n\\tv_pk_add_u16 %3, %3, %4\\n\\t\"\n                 \"v_perm_b32 %0, %1, %0, %5\\n\\tv_perm_b32 %1, %3, %2, %5\\n\\tv_perm_b32 %2, %1, %0, %5\\n\\tv_perm_b32 %3, %3, %2, %5\"\n                 : \"+v\"(a), \"+v\"(b), \"+v\"(c), \"+v\"(d) : \"v\"(0x00800080u), \"v\"(0x07050301u));\n    return (intx4){(int)a, (int)b, (int)c, (int)d};\n  };\n  auto read_a = [&](int par, int asub) {")
    s = s.replace("        Xp[t4] = __builtin_shufflevector(ld16(rdA[0] + o), ld16(rdA[1] + o), 0, 1, 2, 3, 4, 5, 6, 7);",
                  "        if (rd_f8) { const intx4 t = ld16(rdA[0] + o); Xp[t4] = __builtin_shufflevector(t, cvt_cost(t), 0, 1, 2, 3, 4, 5, 6, 7); }\n        else Xp[t4] = __builtin_shufflevector(ld16(rdA[0] + o), ld16(rdA[1] + o), 0, 1, 2, 3, 4, 5, 6, 7);")
    s = s.replace("        Wp[c2] = __builtin_shufflevector(ld16(rdB[0] + o), ld16(rdB[1] + o), 0, 1, 2, 3, 4, 5, 6, 7);",
                  "        if (rd_f8) { const intx4 t = ld16(rdB[0] + o); Wp[c2] = __builtin_shufflevector(t, cvt_cost(t), 0, 1, 2, 3, 4, 5, 6, 7); }\n        else Wp[c2] = __builtin_shufflevector(ld16(rdB[0] + o), ld16(rdB[1] + o), 0, 1, 2, 3, 4, 5, 6, 7);")
    s = s.replace("      for (int kt = nk0; kt < nk; kt += 2) two_ktiles(std::true_type{}, kt + 2 >= nk);",
                  "      rd_f8 = true;\n      for (int kt = nk0; kt < nk; kt += 2) two_ktiles(std::true_type{}, kt + 2 >= nk);\n      rd_f8 = false;")
    if mode == "half6":
        s = s.replace("acc[asub * 4 + t4][b * 2 + c2], 0, 0, 0,\n", "acc[asub * 4 + t4][b * 2 + c2], 2, 2, 0,\n")
    return s

def main():
    src = open(os.path.join(ROOT, "memvul_amd/csrc/gemm_pp.h")).read()
    for mode in (sys.argv[1:] or ("half", "half6", "nocorr")):
        d = os.path.join(HERE, "x_" + mode)
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(os.path.join(ROOT, "memvul_amd/csrc"), os.path.join(d, "x/csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        p = patch(src, mode)
        n = sum(1 for a, b in zip(src.splitlines(), p.splitlines()) if a != b) + abs(len(src.splitlines()) - len(p.splitlines()))
        assert n > 0, mode
        open(os.path.join(d, "x/csrc/gemm_pp.h"), "w").write(p)
        out = os.path.join(HERE, f"lib_{mode}.so")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", os.path.join(d, "x/csrc/engine.hip"), "-o", out]
        print(mode, "lines changed ~", n, flush=True)
        subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
        shutil.rmtree(d)
    print("built")

if __name__ == "__main__":
    main()
