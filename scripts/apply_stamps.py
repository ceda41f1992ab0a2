"""Diagnostic build: patch s_memtime stamps into the igemm kernel (never committed in the patched state).
usage: apply_stamps.py; rebuild; OKP_STAMP=1 python scripts/bench_conv.py ... ; git checkout the three files."""
import os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = R + '/object_keypoints_amd/csrc/okp_internal.h'
s = open(p).read()
s = s.replace("  int32_t n_co_tiles;\n", "  int32_t n_co_tiles;\n  unsigned long long* dbg;\n", 1)
open(p, 'w').write(s)
p = R + '/object_keypoints_amd/csrc/okp_api.hip'
s = open(p).read()
s = s.replace("  return okp_launch_igemm(plan, p, a->tile, (hipStream_t)stream);", """  static unsigned long long* dbg_dev = nullptr;
  static const bool stamp = getenv("OKP_STAMP") != nullptr;
  if (stamp && !dbg_dev) (void)hipMalloc((void**)&dbg_dev, 4096 * 16 * 8);
  p.dbg = stamp ? dbg_dev : nullptr;
  int rc = okp_launch_igemm(plan, p, a->tile, (hipStream_t)stream);
  if (stamp) {
    (void)hipDeviceSynchronize();
    static unsigned long long h[4096 * 16];
    (void)hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost);
    double sum[16] = {0}; int nb = 256;
    for (int b = 0; b < nb; ++b) for (int k = 1; k < 12; ++k) sum[k] += (double)(h[b * 16 + k] - h[b * 16 + k - 1]);
    fprintf(stderr, "stamps (avg cycles over %d WGs, first tile):", nb);
    for (int k = 1; k < 12; ++k) fprintf(stderr, " %d:%.0f", k, sum[k] / nb);
    fprintf(stderr, "\\n");
  }
  return rc;""")
s = s.replace("#include <cstring>\n", "#include <cstring>\n#include <cstdlib>\n", 1)
open(p, 'w').write(s)
p = R + '/object_keypoints_amd/csrc/okp_igemm.hip'
s = open(p).read()
s = s.replace('''__device__ __forceinline__ int fastdiv(''', '''#define STAMP(k) do { if (p.dbg && tid == 0 && slot == (int)blockIdx.x) { unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.dbg[blockIdx.x * 16 + (k)] = t_; } } while (0)

__device__ __forceinline__ int fastdiv(''')
s = s.replace("  const int cls = tile / p.tiles_per_class;", "  STAMP(0);\n  const int cls = tile / p.tiles_per_class;")
s = s.replace("  acc_t acc[TCO][TPX];", "  STAMP(1);\n  acc_t acc[TCO][TPX];")
s = s.replace("  int st_c = 0, st_i = NS - 1; ", "  STAMP(2);\n  int st_c = 0, st_i = NS - 1; ")
s = s.replace("  __syncthreads();                               // all waves done with the last stage before it is reused", "  STAMP(3);\n  __syncthreads();                               // all waves done with the last stage before it is reused\n  STAMP(4);")
s = s.replace("    __syncthreads();\n    constexpr int GROUPS = BCO / 8;", "    if (pass == 0) STAMP(5);\n    __syncthreads();\n    if (pass == 0) STAMP(6);\n    constexpr int GROUPS = BCO / 8;")
s = s.replace("    __syncthreads();       // staging is free again (next pass, or the next tile's LDS-DMA)", "    if (pass == 0) STAMP(7);\n    __syncthreads();       // staging is free again (next pass, or the next tile's LDS-DMA)\n    if (pass == 0) STAMP(8); else STAMP(9);")
s = s.replace("  }  // tile loop", "  STAMP(10);\n  }  // tile loop\n  if (p.dbg && tid == 0) { unsigned long long t_; asm volatile(\"s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)\" : \"=s\"(t_) :: \"memory\"); p.dbg[blockIdx.x * 16 + 11] = t_; }")
open(p, 'w').write(s)
