"""Generates scripts/exp/gemm_f16x3_dbg.hip from csrc/gemm_f16x3.hip: same kernels + per-workgroup timestamps."""
import os, re
here = os.path.dirname(os.path.abspath(__file__))
s = open(os.path.join(here, "../../point_sam_amd/csrc/gemm_f16x3.hip")).read()
assert "void gemm_f16x3_kernel(const F16x3Args p) {" in s
s = s.replace("void gemm_f16x3_kernel(const F16x3Args p) {",
              "void gemm_f16x3_kernel(const F16x3Args p, unsigned long long* dbg) {\n"
              "    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter(); unsigned long long c1 = 0, c2 = 0;")
assert "void gemm_f16x3_pipe_kernel(const F16x3Args p) {" in s
s = s.replace("void gemm_f16x3_pipe_kernel(const F16x3Args p) {",
              "void gemm_f16x3_pipe_kernel(const F16x3Args p, unsigned long long* dbg) {\n"
              "    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_readcyclecounter(); unsigned long long c1 = 0, c2 = 0;")
s = s.replace("    int t = 0;\n    for (; t + 1 < nslabs; t += 2) {   // straight-line pairs", "    c1 = __builtin_readcyclecounter();\n    int t = 0;\n    for (; t + 1 < nslabs; t += 2) {   // straight-line pairs")
s = s.replace("#undef HP_TERM\n", "#undef HP_TERM\n    c2 = __builtin_readcyclecounter();\n")
s = s.replace("    for (int t = 0; t < nslabs; ++t) {\n        __syncthreads();\n        store_split();", "    c1 = __builtin_readcyclecounter();\n    for (int t = 0; t < nslabs; ++t) {\n        __syncthreads();\n        store_split();")
s = s.replace("    // ---- epilogue: un-scale", "    c2 = __builtin_readcyclecounter();\n    // ---- epilogue: un-scale")
s = s.replace("                                  n0 + wn * TN * 32, lane, p.C, p.residual);\n}",
              "                                  n0 + wn * TN * 32, lane, p.C, p.residual);\n"
              "    if (threadIdx.x == 0) { unsigned long long* d = dbg + (size_t)blockIdx.x * 8; const unsigned long long c3 = __builtin_readcyclecounter();\n"
              "        d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime(); d[2] = c1 - c0; d[3] = c2 - c1; d[4] = c3 - c2;\n"
              "        d[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4); d[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20); d[7] = tile; }\n}")
s = re.sub(r"hipLaunchKernelGGL\(\((gemm_f16x3\w*kernel)<([^>]*)>\), grid, dim3\(256\), 0, stream, p\)", r"hipLaunchKernelGGL((\1<\2>), grid, dim3(256), 0, stream, p, g_dbg)", s)
s = s.replace("static int g_f16x3_cfg = -1;", "static unsigned long long* g_dbg = nullptr;\nPSAM_API void dbg_set_buffer(unsigned long long* b) { g_dbg = b; }\nstatic int g_f16x3_cfg = -1;")
s += '\nvoid psam_set_error(const char*) {}\n'
open(os.path.join(here, "gemm_f16x3_dbg.hip"), "w").write(s)
