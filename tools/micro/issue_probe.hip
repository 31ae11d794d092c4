// Diagnostics: what one wave pays per instruction on gfx950 (shader-clock ticks via s_memtime).
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/issue_probe tools/micro/issue_probe.hip && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__device__ inline unsigned long long now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// every test: 64 copies of the pattern between two stamps, 32 rounds; out[test] = ticks per copy
__global__ __launch_bounds__(64) void k_probe(unsigned long long* out, uint32_t seed, const uint32_t* gmem) {
  __shared__ uint32_t lds[256];
  lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 64] = seed;
  __syncthreads();
  uint32_t s = seed, s2 = seed + 1, v = threadIdx.x + seed, v2 = threadIdx.x;
  unsigned long long t0, t1;
  int ti = 0;
  auto rec = [&](unsigned long long d) { if (threadIdx.x == 0) out[blockIdx.x * 32 + ti] = d; ++ti; };
  // 0: empty (stamp overhead)
  t0 = now(); t1 = now(); rec(t1 - t0);
  // 1: dependent s_add_u32 (4-byte instr)
  t0 = now(); REP64(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s) : "s"(s2));) t1 = now(); rec(t1 - t0);
  // 2: dependent s_mul_i32 with literal (8-byte instr)
  t0 = now(); REP64(asm volatile("s_mul_i32 %0, %0, 0x9e3779b1" : "+s"(s));) t1 = now(); rec(t1 - t0);
  // 3: independent s_add pairs
  t0 = now(); REP64(asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1" : "+s"(s), "+s"(s2));) t1 = now(); rec(t1 - t0);
  // 4: dependent v_add_u32
  t0 = now(); REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(v2));) t1 = now(); rec(t1 - t0);
  // 5: dependent v_mul_lo_u32
  t0 = now(); REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v) : "v"(v2));) t1 = now(); rec(t1 - t0);
  // 6: v_readfirstlane -> s_add -> v_mov (vector/scalar ping-pong)
  t0 = now(); REP64(asm volatile("v_readfirstlane_b32 %1, %0\n\ts_add_u32 %1, %1, 1\n\tv_mov_b32 %0, %1" : "+v"(v), "+s"(s));) t1 = now(); rec(t1 - t0);
  // 7: s_cmp + taken forward branch over one instruction
  t0 = now(); REP64(asm volatile("s_cmp_eq_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n1:" : "+s"(s));) t1 = now(); rec(t1 - t0);
  // 8: s_cmp + not-taken branch
  t0 = now(); REP64(asm volatile("s_cmp_lg_u32 %0, %0\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n1:" : "+s"(s));) t1 = now(); rec(t1 - t0);
  // 9: LDS read + wait (dependent address)
  t0 = now(); REP64(asm volatile("ds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_and_b32 %0, 0xfc, %0" : "+v"(v2));) t1 = now(); rec(t1 - t0);
  // 10: global load + wait, same address (L1/L2 hit)
  { const uint32_t* p = gmem; uint32_t r;
    t0 = now(); REP64(asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");) t1 = now(); rec(t1 - t0); v ^= r; }
  // 11: v_cmp + s_cbranch_vccnz not taken
  t0 = now(); REP64(asm volatile("v_cmp_eq_u32 vcc, 0x7fffffff, %0\n\ts_cbranch_vccnz 1f\n\ts_nop 0\n1:" : : "v"(v2) : "vcc");) t1 = now(); rec(t1 - t0);
  // 12: v_lshrrev_b64 dependent
  { unsigned long long q = v;
    t0 = now(); REP64(asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(q));) t1 = now(); rec(t1 - t0); v ^= (uint32_t)q; }
  // 13: v_readlane with SGPR index then s use
  t0 = now(); REP64(asm volatile("s_and_b32 %1, %1, 63\n\ts_nop 3\n\tv_readlane_b32 %1, %0, %1" : "+v"(v), "+s"(s));) t1 = now(); rec(t1 - t0);
  // 14: s_nop 0 x1 (pure issue)
  t0 = now(); REP64(asm volatile("s_nop 0");) t1 = now(); rec(t1 - t0);
  // 15: dependent f64 mul
  { double d = (double)v;
    t0 = now(); REP64(asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d));) t1 = now(); rec(t1 - t0); v ^= (uint32_t)d; }
  if (threadIdx.x == 0) out[blockIdx.x * 32 + 31] = s + s2 + v + v2;
}

int main() {
  const char* names[] = {"stamp pair", "s_add dep", "s_mul literal dep", "2 x s_add indep", "v_add dep", "v_mul_lo dep",
                         "readfirstlane+s_add+v_mov", "s_cmp+taken branch", "s_cmp+untaken branch+s_add", "ds_read+wait+v_and",
                         "global_load+wait (hit)", "v_cmp+vccnz untaken+s_nop", "v_lshrrev_b64 dep", "s_and+s_nop3+v_readlane",
                         "s_nop 0", "v_mul_f64 dep"};
  uint32_t* g; hipMalloc(&g, 4096); hipMemset(g, 0, 4096);
  for (int blocks : {1, 256 * 4, 256 * 12, 256 * 24}) {
    unsigned long long* d; hipMalloc(&d, blocks * 32 * 8);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(64), 0, 0, d, 12345u, g);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 32);
    hipMemcpy(h.data(), d, blocks * 32 * 8, hipMemcpyDeviceToHost);
    printf("---- %d single-wave workgroups (%.1f per SIMD)\n", blocks, blocks / 1024.0);
    for (int t = 0; t < 16; ++t) {
      double sum = 0; for (int b = 0; b < blocks; ++b) sum += (double)h[b * 32 + t];
      const double per = sum / blocks, base = 0;
      printf("%-30s %8.1f ticks per 64 copies -> %6.2f per copy\n", names[t], per, t ? per / 64.0 : per);
      (void)base;
    }
    hipFree(d);
  }
  return 0;
}
