"""Timing-only knock-out variants of the fused SA kernels (WRONG results by construction, never loaded by the product): which part of
the non-matrix fifth of sa1_fused_kernel / sa2_fused_kernel is what.  Copies wave_mlp.h / sa_fused.hip into scratch/knock/<variant>/, patches
them, links the object with the product's other objects into scratch/knock/lib_<variant>.so; run with
    ANCSH_HIP_LIB=$PWD/scratch/knock/lib_<variant>.so python tools/sa_steady.py 1000
Variants (combine with '_'): base | novalu (hidden layers' epilogues store the raw accumulators: no packed add / fma / max) | nogather (no
index / coordinate / feature loads) | wl1 (every weight load reads the same L1-resident lines).  profiles/r05_sa_knockout.txt.
(Two more were tried and are not here because the compiler changed more than the knocked-out part: dropping the in-place LDS stores keeps
both layers' maxima alive and raises the register pressure; dropping the pooled epilogue lets it delete half the MFMAs.)"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "articulated-pose_amd", "csrc")
HERE = os.path.join(ROOT, "scratch", "knock")
os.makedirs(HERE, exist_ok=True)

def rep(s, a, b, count=1):
    assert a in s, a[:70]
    return s.replace(a, b) if count == 0 else s.replace(a, b, count)

def variant(name):
    d = os.path.join(HERE, name)
    os.makedirs(d, exist_ok=True)
    for f in ("wave_mlp.h", "common.h", "sa_fused.hip"):
        shutil.copy(os.path.join(SRC, f), d)
    h = open(os.path.join(d, "wave_mlp.h")).read()
    k = open(os.path.join(d, "sa_fused.hip")).read()
    k = rep(k, '#include "../../include/ancsh_hip.h"', '#include "%s/include/ancsh_hip.h"' % ROOT) if '../../include' in k else k
    h = rep(h, '#include "../../include/ancsh_hip.h"', '#include "%s/include/ancsh_hip.h"' % ROOT) if '../../include' in h else h
    if "novalu" in name:
        h = rep(h, """                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                a = __builtin_elementwise_fma(a + b2, s2, t2);
                if (POOL) {""", """                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                if (POOL) a = __builtin_elementwise_fma(a + b2, s2, t2);
                if (POOL) {""")
        h = rep(h, "const float v0 = RELU ? fmaxf(a.x, 0.f) : a.x, v1 = RELU ? fmaxf(a.y, 0.f) : a.y;\n                    const int row = i * 32", "const float v0 = a.x, v1 = a.y;\n                    const int row = i * 32")
    if "spread" in name:        # candidate, not a knock-out: the slot's weight loads one per column tile, each right behind that tile's MFMAs
        h = rep(h, """        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < RT; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[s % (DA + 1)][i], f4_get(bw[slot % (DW + 1)][j], q), acc[i][j], 0, 0, 0);
        if (q == 0) w_load<K, N, WSTRIDE>(L, bw[(slot + DW) % (DW + 1)], slot + DW);""", """        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[s % (DA + 1)][i], f4_get(bw[slot % (DW + 1)][j], q), acc[i][j], 0, 0, 0);
            if (q == 0 && slot + DW < C::NS) {
                const float4 *Wp = reinterpret_cast<const float4 *>(L.w) + (threadIdx.x & 63);
                bw[(slot + DW) % (DW + 1)][j] = WSTRIDE ? Wp[(size_t)(slot + DW) * L.wstride + j * 64] : Wp[(size_t)((slot + DW) * TN + j) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }""")
    if "prioep" in name:        # candidate: a wave raises its issue priority for its epilogues (and, prioepg, its gather), drops it for the k loops
        h = rep(h, "    wave_lds_fence();     // every read of T by the k loop has completed (its value fed an MFMA already issued)\n#pragma unroll\n    for (int j = 0; j < N / 32; ++j) {\n        const int col = j * 32 + l31;\n        float m = 0.f;",
                   "    __builtin_amdgcn_s_setprio(3);\n    wave_lds_fence();\n#pragma unroll\n    for (int j = 0; j < N / 32; ++j) {\n        const int col = j * 32 + l31;\n        float m = 0.f;")
        h = rep(h, "        if (POOL) pm[j] = fmaxf(m, __shfl_xor(m, 32, 64));\n    }\n}", "        if (POOL) pm[j] = fmaxf(m, __shfl_xor(m, 32, 64));\n    }\n    __builtin_amdgcn_s_setprio(0);\n}")
        if "prioepg" in name:
            k = rep(k, "    float4 bw1[LayerCfg<CIN, C1>::DW + 1][C1 / 32];\n    w_prologue<CIN, C1>(L1, bw1);", "    __builtin_amdgcn_s_setprio(3);\n    float4 bw1[LayerCfg<CIN, C1>::DW + 1][C1 / 32];\n    w_prologue<CIN, C1>(L1, bw1);")
            k = rep(k, "    __builtin_amdgcn_sched_barrier(0);\n    SA_STAMP(0);", "    __builtin_amdgcn_s_setprio(0);\n    __builtin_amdgcn_sched_barrier(0);\n    SA_STAMP(0);")
    if "prioloop" in name:      # the opposite: the k loops at high priority
        h = rep(h, "    wave_lds_fence();                             // the tile (gather or the previous layer's epilogue) is complete", "    __builtin_amdgcn_s_setprio(3);\n    wave_lds_fence();")
        h = rep(h, "        if (s == EP_AT) ep_load<N>(L, ep);\n        __builtin_amdgcn_sched_barrier(0);\n    }\n}", "        if (s == EP_AT) ep_load<N>(L, ep);\n        __builtin_amdgcn_sched_barrier(0);\n    }\n    __builtin_amdgcn_s_setprio(0);\n}")
    if "noepi" in name:         # knock-out: the hidden layers' epilogues are skipped at run time (a scalar branch on n < 0): the bound of ANY epilogue optimisation
        k = rep(k, "        epilogue<C1, LD, false, RT>(T, acc, ep1, none1);", "        if (n < 0) epilogue<C1, LD, false, RT>(T, acc, ep1, none1);")
        k = rep(k, "        epilogue<C2, LD, false, RT>(T, acc2, ep2, none2);", "        if (n < 0) epilogue<C2, LD, false, RT>(T, acc2, ep2, none2);")
    if "nopoolep" in name:      # knock-out: the pooled epilogue of layer 3 reduced to a plain max of the raw accumulators (no packed add / fma)
        h = rep(h, "                a = __builtin_elementwise_fma(a + b2, s2, t2);\n                if (POOL) {", "                if (!POOL) a = __builtin_elementwise_fma(a + b2, s2, t2);\n                if (POOL) {")
    if "poolmm" in name:        # candidate (exact): pooled epilogue as max AND min of the raw accumulators, bias + BN + ReLU once per column on the extremum the sign of scale selects
        h = rep(h, """        float m = 0.f;    // post-ReLU values are >= 0""", """        float m = 0.f;    // post-ReLU values are >= 0
        float mx = -__builtin_inff(), mn = __builtin_inff();""")
        h = rep(h, """                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                a = __builtin_elementwise_fma(a + b2, s2, t2);
                if (POOL) {
                    m = fmaxf(fmaxf(m, a.x), a.y);            // one v_max3_f32: the running maximum starts at 0, so the ReLU is implicit
                } else {""", """                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                if (POOL) {
                    mx = fmaxf(fmaxf(mx, a.x), a.y);          // v_max3_f32 / v_min3_f32 on the RAW accumulators
                    mn = fminf(fminf(mn, a.x), a.y);
                } else {
                    a = __builtin_elementwise_fma(a + b2, s2, t2);""")
        h = rep(h, """        if (POOL) pm[j] = fmaxf(m, __shfl_xor(m, 32, 64));""", """        if (POOL) {
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            mn = fminf(mn, __shfl_xor(mn, 32, 64));
            const float sc = ep[1][j], x = sc < 0.f ? mn : mx;
            const float v = sc == 0.f ? ep[2][j] : __builtin_fmaf(x + ep[0][j], sc, ep[2][j]);
            pm[j] = fmaxf(v, m);
        }""")
    if "poolmax" in name:       # candidate: pooled epilogue = running max of the raw accumulators, bias + BN + ReLU once per column at the end (exact when scale >= 0)
        h = rep(h, """                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                a = __builtin_elementwise_fma(a + b2, s2, t2);
                if (POOL) {
                    m = fmaxf(fmaxf(m, a.x), a.y);            // one v_max3_f32: the running maximum starts at 0, so the ReLU is implicit
                } else {""", """                ep_f2 a = {acc[i][j][r], acc[i][j][r + 1]};
                if (POOL) {
                    if (i == 0 && r == 0) m = fmaxf(a.x, a.y); else m = fmaxf(fmaxf(m, a.x), a.y);
                } else {
                    a = __builtin_elementwise_fma(a + b2, s2, t2);""")
        h = rep(h, """        if (POOL) pm[j] = fmaxf(m, __shfl_xor(m, 32, 64));""", """        if (POOL) {
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            pm[j] = fmaxf(__builtin_fmaf(m + ep[0][j], ep[1][j], ep[2][j]), 0.f);
        }""")
    if "nogather" in name:
        k = rep(k, "    if (live) {\n        const long b = cloud;", "    if (live && n < 0) {\n        const long b = cloud;")
        k = rep(k, "    } else {\n        for (int e = lane; e < ROWS * LD; e += 64) T[e] = 0.f;", "    } else if (!live) {\n        for (int e = lane; e < ROWS * LD; e += 64) T[e] = 0.f;")
    if "wl1" in name:           # every weight load hits the same (L1-resident) lines
        h = rep(h, "b[j] = WSTRIDE ? Wp[(size_t)slot * L.wstride + j * 64] : Wp[(size_t)(slot * TN + j) * 64];", "b[j] = Wp[(size_t)j * 64];")
    open(os.path.join(d, "wave_mlp.h"), "w").write(h)
    open(os.path.join(d, "sa_fused.hip"), "w").write(k)
    obj = os.path.join(d, "sa_fused.o")
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -mllvm -pragma-unroll-threshold=4000000 -mllvm -unroll-threshold=4000000".split()
    # candidates that need tools/experiments/sa_wpb_nb.patch applied to csrc/sa_fused.hip first (git apply; measured, not kept):
    # waves per workgroup (wpb<sa1><sa2>, e.g. wpb12 = sa1_fused_kernel in 1-wave workgroups, sa2_fused_kernel in 2-wave ones)
    defs = []
    for part in name.split("_"):
        if part.startswith("wpb") and len(part) == 5:
            defs += ["SA1_WPB=" + part[3], "SA2_WPB=" + part[4]]
        if part.startswith("nb") and part[2:].isdigit():          # nb<k>: k neighbourhood tiles per workgroup, one after the other
            defs += ["SA_NB=" + part[2:]]
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-D" + x for x in defs] + ["-I", SRC, "-c", os.path.join(d, "sa_fused.hip"), "-o", obj])
    others = [os.path.join(SRC, "build", o) for o in os.listdir(os.path.join(SRC, "build")) if o.endswith(".o") and o != "sa_fused.o" and "stamps" not in o]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + others + [obj, "-o", os.path.join(HERE, "lib_%s.so" % name)])
    print("built", name)

for v in sys.argv[1:]:
    variant(v)
