// dropin_host.cpp — TEST-ONLY driver that RUNS the C++ drop-in class (kangaroo_amd/dropin/GPUEngine_amd.cpp) the
// way Kangaroo::SolveKeyGPU does (core/kangaroo.cpp:408-500): constructor, SetWildOffset, SetParams, SetKangaroos,
// callKernel, Launch x L with a SetKangaroo in between, GetKangaroos.  Compiled against the stand-in headers of
// tests/dropin_stub (the reference tree is absent on the GPU box).  Input and output are flat binary files of
// uint64 written / read by tests/test_gpu_dropin.py, which compares the output with the oracle.
//
// input : groups, perGroup, maxFound, dpMask, launches, pokeAfterLaunch, pokeIdx,
//         wildOffset[4], jd[64][4], jx[64][4], jy[64][4], px[n][4], py[n][4], d[n][4], poke x[4] y[4] d[4]
// output: n, memoryBytes, nbThread, groupSize, per launch: count, then count x (x[4], d[4], kIdx); then herd px py d
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "engine.h"

static std::vector<uint64_t> read_all(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) {
    perror(path);
    exit(2);
  }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<uint64_t> v(sz / 8);
  if (fread(v.data(), 8, v.size(), f) != v.size()) exit(2);
  fclose(f);
  return v;
}

static void load_ints(std::vector<Int> &dst, const uint64_t *src, size_t n) {
  dst.resize(n);
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) dst[i].bits64[k] = src[4 * i + k];
    dst[i].bits64[4] = 0;
  }
}

int main(int argc, char **argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: dropin_host <case.bin> <out.bin>\n");
    return 2;
  }
  std::vector<uint64_t> in = read_all(argv[1]);
  const uint64_t *p = in.data();
  const int groups = (int)*p++, per_group = (int)*p++;
  const uint32_t max_found = (uint32_t)*p++;
  const uint64_t dp_mask = *p++;
  const int launches = (int)*p++, poke_after = (int)*p++;
  const uint64_t poke_idx = *p++;
  const uint64_t n = (uint64_t)GPU_GRP_SIZE * groups * per_group;

  int gx = 0, gy = 0;
  if (!GPUEngine::GetGridSize(0, &gx, &gy) || gx <= 0 || gy <= 0) {  // main thread, before any engine (core/kangaroo.cpp:712)
    fprintf(stderr, "GetGridSize failed\n");
    return 3;
  }
  GPUEngine *gpu = new GPUEngine(groups, per_group, 0, max_found);
  if (gpu->GetNbThread() != groups * per_group || gpu->GetGroupSize() != GPU_GRP_SIZE || gpu->deviceName.empty()) {
    fprintf(stderr, "engine geometry / name wrong\n");
    return 3;
  }
  std::vector<Int> wild, jd, jx, jy, px, py, d, poke;
  load_ints(wild, p, 1);
  p += 4;
  load_ints(jd, p, NB_JUMP);
  p += 4 * NB_JUMP;
  load_ints(jx, p, NB_JUMP);
  p += 4 * NB_JUMP;
  load_ints(jy, p, NB_JUMP);
  p += 4 * NB_JUMP;
  load_ints(px, p, n);
  p += 4 * n;
  load_ints(py, p, n);
  p += 4 * n;
  load_ints(d, p, n);
  p += 4 * n;
  load_ints(poke, p, 3);

  gpu->SetWildOffset(&wild[0]);
  gpu->SetParams(dp_mask, jd.data(), jx.data(), jy.data());
  gpu->SetKangaroos(px.data(), py.data(), d.data());
  if (!gpu->callKernel()) return 4;

  std::vector<uint64_t> out;
  out.push_back(n);
  out.push_back((uint64_t)gpu->GetMemory());
  out.push_back((uint64_t)gpu->GetNbThread());
  out.push_back((uint64_t)gpu->GetGroupSize());
  std::vector<ITEM> found;
  for (int l = 0; l < launches; ++l) {
    if (!gpu->Launch(found)) return 5;  // waits for launch l, queues launch l + 1
    out.push_back(found.size());
    for (const ITEM &it : found) {
      for (int k = 0; k < 4; ++k) out.push_back(it.x.bits64[k]);
      for (int k = 0; k < 4; ++k) out.push_back(it.d.bits64[k]);
      out.push_back(it.kIdx);
      if (it.x.bits64[4] || it.d.bits64[4]) return 6;  // limb 4 must be zero (engine.cu:528-533)
    }
    if (l == poke_after) gpu->SetKangaroo(poke_idx, &poke[0], &poke[1], &poke[2]);  // core/kangaroo.cpp:484-485
  }
  // one launch is still in flight, as at a save request (core/kangaroo.cpp:495-496): GetKangaroos waits for it
  gpu->GetKangaroos(px.data(), py.data(), d.data());
  for (const std::vector<Int> *v : {&px, &py, &d})
    for (uint64_t i = 0; i < n; ++i) {
      for (int k = 0; k < 4; ++k) out.push_back((*v)[i].bits64[k]);
      if ((*v)[i].bits64[4]) return 6;
    }
  delete gpu;
  FILE *f = fopen(argv[2], "wb");
  if (!f || fwrite(out.data(), 8, out.size(), f) != out.size()) return 7;
  fclose(f);
  return 0;
}
