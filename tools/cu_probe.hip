// Which compute units does a CU-masked HIP stream reach?  For each mask pattern: launch many blocks on a stream created by
// hipExtStreamCreateWithCUMask and count the distinct (XCC, SE, CU) triples the blocks report (s_getreg HW_ID / XCC_ID).
// build: hipcc --offload-arch=gfx950 -O2 tools/cu_probe.hip -o tools/cu_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>

__global__ void k_probe(unsigned* out) {
  // busy-wait a little so that blocks spread over every CU the stream may use
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID (id 4), all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID (id 20)
    out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
  }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int nb = 4096;
  unsigned* d; hipMalloc(&d, nb * 2 * sizeof(unsigned));
  hipLaunchKernelGGL(k_probe, dim3(nb), dim3(256), 0, s, d);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(nb * 2);
  hipMemcpy(h.data(), d, nb * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::set<unsigned> cus; std::map<unsigned, std::set<unsigned>> per_xcc;
  for (int i = 0; i < nb; i++) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    unsigned id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    cus.insert(id); per_xcc[xcc].insert(id & 0xfff);
  }
  printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
  for (auto& kv : per_xcc) printf(" x%u:%zu", kv.first, kv.second.size());
  printf("\n");
  hipFree(d); hipStreamDestroy(s);
}

int main() {
  auto bits = [](int lo, int hi, int words) { std::vector<uint32_t> m(words, 0); for (int b = lo; b < hi; b++) m[b >> 5] |= 1u << (b & 31); return m; };
  run("all 256 (8 words)", bits(0, 256, 8));
  run("bits 0..31", bits(0, 32, 8));
  run("bits 0..31 (1 word)", bits(0, 32, 1));
  run("bits 32..255", bits(32, 256, 8));
  run("bits 0..7", bits(0, 8, 8));
  run("bits 0..63", bits(0, 64, 8));
  run("bits 64..255", bits(64, 256, 8));
  run("bits 224..255", bits(224, 256, 8));
  run("bits 0..127", bits(0, 128, 8));
  { std::vector<uint32_t> m(8, 0); for (int b = 0; b < 256; b += 8) m[b >> 5] |= 1u << (b & 31); run("every 8th bit", m); }
  { std::vector<uint32_t> m(8, 0xFFFFFFF0u); run("all but low 4 bits of each word", m); }
  { std::vector<uint32_t> m(8, 0x0000000Fu); run("low 4 bits of each word", m); }
  return 0;
}
