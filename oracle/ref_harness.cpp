// ref_harness.cpp — golden-vector generator.  TEST INFRASTRUCTURE ONLY.
//
// Own code; it is compiled TOGETHER WITH the reference's CPU translation units where
// they lie under /root/reference (oracle/Makefile, output oracle/_ref/ref_harness) and
// calls the reference's own Int / IntGroup / Secp256K1 / Kangaroo code to produce the
// JSON fixtures committed under tests/golden/.  No reference source is copied.
//
// The walk fixture drives the real Kangaroo::SolveKeyCPU (core/kangaroo.cpp:258-395)
// on a small known-key range and dumps the DP hash table it leaves behind.
//
// Usage: ref_harness <out.json>
//        ref_harness --savework <out.bin> <out.json>   the reference's own Kangaroo::SaveWork writes a work file
//        ref_harness --loadwork <in.bin> <out.json>    the reference's own LoadWork + FetchWalks read one
// standard headers first, so that the access override below only touches kangaroo.h
#include <algorithm>
#include <array>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>
#define private public  // reach Kangaroo::CreateHerd/CreateJumpTable/hashTable (test only)
#include "kangaroo.h"
#undef private
#include "engine.h"

#include <cinttypes>

// ---- link stubs: the CPU-only harness never constructs a GPU engine ----------------
GPUEngine::GPUEngine(int, int, int, uint32_t) { initialised = false; }
GPUEngine::~GPUEngine() {}
void GPUEngine::SetParams(uint64_t, Int *, Int *, Int *) {}
void GPUEngine::SetKangaroos(Int *, Int *, Int *) {}
void GPUEngine::GetKangaroos(Int *, Int *, Int *) {}
void GPUEngine::SetKangaroo(uint64_t, Int *, Int *, Int *) {}
bool GPUEngine::Launch(std::vector<ITEM> &, bool) { return false; }
void GPUEngine::SetWildOffset(Int *) {}
int GPUEngine::GetNbThread() { return 0; }
int GPUEngine::GetGroupSize() { return GPU_GRP_SIZE; }
int GPUEngine::GetMemory() { return 0; }
bool GPUEngine::callKernelAndWait() { return false; }
bool GPUEngine::callKernel() { return false; }
void *GPUEngine::AllocatePinnedMemory(size_t) { return nullptr; }
void GPUEngine::FreePinnedMemory(void *) {}
void GPUEngine::PrintCudaInfo() {}
bool GPUEngine::GetGridSize(int, int *, int *) { return false; }

static std::string hx(Int &a) {
  char b[80];
  snprintf(b, sizeof b, "%016" PRIx64 "%016" PRIx64 "%016" PRIx64 "%016" PRIx64, a.bits64[3],
           a.bits64[2], a.bits64[1], a.bits64[0]);
  return b;
}
static std::string hx4(const uint64_t *v) {
  char b[80];
  snprintf(b, sizeof b, "%016" PRIx64 "%016" PRIx64 "%016" PRIx64 "%016" PRIx64, v[3], v[2], v[1],
           v[0]);
  return b;
}
static Int fromLimbs(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3) {
  Int r;
  r.SetInt32(0);
  r.bits64[0] = a0;
  r.bits64[1] = a1;
  r.bits64[2] = a2;
  r.bits64[3] = a3;
  r.bits64[4] = 0;
  return r;
}

static void fieldKats(FILE *f) {
  std::vector<Int> vals;
  const uint64_t F = 0xFFFFFFFFFFFFFFFFULL;
  // edge values: 0,1,2,p-1,p,p+1,2^256-1, 2^255, K-1, K, (p+1)/2, small, high-limb-only
  vals.push_back(fromLimbs(0, 0, 0, 0));
  vals.push_back(fromLimbs(1, 0, 0, 0));
  vals.push_back(fromLimbs(2, 0, 0, 0));
  vals.push_back(fromLimbs(0xFFFFFFFEFFFFFC2EULL, F, F, F));
  vals.push_back(fromLimbs(0xFFFFFFFEFFFFFC2FULL, F, F, F));
  vals.push_back(fromLimbs(0xFFFFFFFEFFFFFC30ULL, F, F, F));
  vals.push_back(fromLimbs(F, F, F, F));
  vals.push_back(fromLimbs(0, 0, 0, 0x8000000000000000ULL));
  vals.push_back(fromLimbs(0x1000003D0ULL, 0, 0, 0));
  vals.push_back(fromLimbs(0x1000003D1ULL, 0, 0, 0));
  vals.push_back(fromLimbs(0xFFFFFFFF7FFFFE18ULL, F, F, 0x7FFFFFFFFFFFFFFFULL));  // (p+1)/2
  vals.push_back(fromLimbs(0, 0, 0, 1));
  vals.push_back(fromLimbs(F, 0, F, 0));
  vals.push_back(fromLimbs(0, F, 0, F));
  vals.push_back(fromLimbs(0xFFFFFFFFULL, 0xFFFFFFFF00000000ULL, 0xFFFFFFFFULL, 0xFFFFFFFF00000000ULL));
  rseed(0xC0FFEE);
  for (int i = 0; i < 49; i++) {
    Int r;
    r.Rand(256);
    vals.push_back(r);
  }
  fprintf(f, "\"field\":[\n");
  bool first = true;
  for (size_t i = 0; i < vals.size(); i++) {
    for (size_t j = 0; j < vals.size(); j++) {
      // all pairs among the 15 edge values, plus a diagonal band for the random ones
      if (!(i < 15 && j < 15) && !(j == (i * 7 + 3) % vals.size()) && !(i >= 15 && j == i)) continue;
      Int a(&vals[i]), b(&vals[j]);
      Int m, s, d, inv;
      m.ModMulK1(&a, &b);
      s.ModSquareK1(&a);
      d.ModSub(&a, &b);
      inv.Set(&a);
      inv.ModInv();
      fprintf(f, "%s{\"a\":\"%s\",\"b\":\"%s\",\"mul\":\"%s\",\"sqr\":\"%s\",\"sub\":\"%s\",\"inv\":\"%s\"}",
              first ? "" : ",\n", hx(a).c_str(), hx(b).c_str(), hx(m).c_str(), hx(s).c_str(),
              hx(d).c_str(), hx(inv).c_str());
      first = false;
    }
  }
  fprintf(f, "],\n");
}

static void orderKats(FILE *f, Secp256K1 &secp) {
  rseed(0xBADC0DE);
  fprintf(f, "\"order\":[\n");
  for (int i = 0; i < 24; i++) {
    Int a, b;
    a.Rand(i < 8 ? 256 : 200);
    b.Rand(i < 16 ? 256 : 130);
    if (i < 8) {  // bring into [0,n)
      a.Mod(&secp.order);
      b.Mod(&secp.order);
    }
    Int s(&a), d(&a), n(&a);
    s.ModAddK1order(&b);
    d.ModSubK1order(&b);
    n.ModNegK1order();
    fprintf(f, "%s{\"a\":\"%s\",\"b\":\"%s\",\"add\":\"%s\",\"sub\":\"%s\",\"neg\":\"%s\"}", i ? ",\n" : "",
            hx(a).c_str(), hx(b).c_str(), hx(s).c_str(), hx(d).c_str(), hx(n).c_str());
  }
  fprintf(f, "],\n");
}

static void groupInvKat(FILE *f) {
  rseed(0x1234ABCD);
  const int n = 40;
  std::vector<Int> v(n), w(n);
  for (int i = 0; i < n; i++) {
    v[i].Rand(256);
    w[i].Set(&v[i]);
  }
  IntGroup grp(n);
  grp.Set(w.data());
  grp.ModInv();
  fprintf(f, "\"groupinv\":{\"in\":[");
  for (int i = 0; i < n; i++) fprintf(f, "%s\"%s\"", i ? "," : "", hx(v[i]).c_str());
  fprintf(f, "],\"out\":[");
  for (int i = 0; i < n; i++) fprintf(f, "%s\"%s\"", i ? "," : "", hx(w[i]).c_str());
  fprintf(f, "]},\n");
}

static void rngKats(FILE *f) {
  fprintf(f, "\"rng\":{");
  rseed(12345);
  fprintf(f, "\"seed\":12345,\"draws\":[");
  for (int i = 0; i < 8; i++) fprintf(f, "%s%lu", i ? "," : "", rndl());
  fprintf(f, "],");
  rseed(12345);
  Int a, b, c;
  a.Rand(64);
  b.Rand(40);
  c.Rand(125);
  fprintf(f, "\"rand64\":\"%s\",\"rand40\":\"%s\",\"rand125\":\"%s\"},\n", hx(a).c_str(),
          hx(b).c_str(), hx(c).c_str());
}

static void ecKats(FILE *f, Secp256K1 &secp) {
  rseed(0x5EC9256);
  fprintf(f, "\"ec\":[\n");
  std::vector<Int> ks;
  for (int i = 0; i < 12; i++) {
    Int k;
    if (i == 0) k.SetInt32(1);
    else if (i == 1) k.SetInt32(2);
    else if (i == 2) k.SetBase16("A5B3C1D2E7");
    else if (i == 3) k.SetBase16("0123456789ABCDEF");
    else k.Rand(i < 8 ? 256 : 64);
    if (i >= 4 && i < 8) k.Mod(&secp.order);
    ks.push_back(k);
  }
  std::vector<Point> pts = secp.ComputePublicKeys(ks);
  for (size_t i = 0; i < ks.size(); i++) {
    Point s = secp.ComputePublicKey(&ks[i]);
    fprintf(f, "%s{\"k\":\"%s\",\"x\":\"%s\",\"y\":\"%s\",\"bx\":\"%s\",\"by\":\"%s\",\"pub\":\"%s\"}",
            i ? ",\n" : "", hx(ks[i]).c_str(), hx(s.x).c_str(), hx(s.y).c_str(), hx(pts[i].x).c_str(),
            hx(pts[i].y).c_str(), secp.GetPublicKeyHex(true, s).c_str());
  }
  fprintf(f, "],\n");
  // affine additions
  fprintf(f, "\"ecadd\":[\n");
  for (size_t i = 0; i + 1 < pts.size(); i++) {
    Point r = secp.AddDirect(pts[i], pts[i + 1]);
    fprintf(f, "%s{\"x1\":\"%s\",\"y1\":\"%s\",\"x2\":\"%s\",\"y2\":\"%s\",\"rx\":\"%s\",\"ry\":\"%s\"}",
            i ? ",\n" : "", hx(pts[i].x).c_str(), hx(pts[i].y).c_str(), hx(pts[i + 1].x).c_str(),
            hx(pts[i + 1].y).c_str(), hx(r.x).c_str(), hx(r.y).c_str());
  }
  fprintf(f, "],\n");
}

struct Ctx {
  Secp256K1 *secp;
  std::string empty;
};

static Kangaroo *makeKangaroo(Ctx &c, int dp) {
  return new Kangaroo(c.secp, dp, false, c.empty, c.empty, 60, false, false, 0.0, 3000, 17403, 3000,
                      "", "/dev/null", false);
}

static void jumpTables(FILE *f, Ctx &c) {
  fprintf(f, "\"jump_tables\":[\n");
  int powers[] = {40, 64, 80, 125};
  for (int pi = 0; pi < 4; pi++) {
    Kangaroo *k = makeKangaroo(c, 8);
    k->rangeStart.SetInt32(0);
    k->rangeEnd.SetInt32(1);
    k->rangeEnd.ShiftL(powers[pi]);
    k->rangeEnd.SubOne();
    k->InitRange();
    k->CreateJumpTable();
    fprintf(f, "%s{\"rangePower\":%d,\"jD\":[", pi ? ",\n" : "", k->rangePower);
    for (int i = 0; i < NB_JUMP; i++) fprintf(f, "%s\"%s\"", i ? "," : "", hx(k->jumpDistance[i]).c_str());
    fprintf(f, "],\"jPx\":[");
    for (int i = 0; i < NB_JUMP; i++) fprintf(f, "%s\"%s\"", i ? "," : "", hx(k->jumpPointx[i]).c_str());
    fprintf(f, "],\"jPy\":[");
    for (int i = 0; i < NB_JUMP; i++) fprintf(f, "%s\"%s\"", i ? "," : "", hx(k->jumpPointy[i]).c_str());
    fprintf(f, "]}");
    delete k;
  }
  fprintf(f, "],\n");
}

// Real SolveKeyCPU on a known key.  grp = CPU_GRP_SIZE used for the run.
static void solveFixture(FILE *f, Ctx &c, const char *name, int rangeBits, const char *startHex,
                         const char *privHex, int dp, int grp, uint32_t seed, bool last) {
  Kangaroo *k = makeKangaroo(c, dp);
  Int start, end, priv;
  start.SetBase16((char *)startHex);
  end.Set(&start);
  Int w;
  w.SetInt32(1);
  w.ShiftL(rangeBits);
  w.SubOne();
  end.Add(&w);
  priv.SetBase16((char *)privHex);
  Point pub = c.secp->ComputePublicKey(&priv);
  if (!k->SetSearchContext(start, end, pub, name)) exit(2);
  k->CreateJumpTable();
  k->SetDP(dp);
  k->CPU_GRP_SIZE = grp;
  k->counters.fill(0);
  k->nbCPUThread = 1;
  k->nbGPUThread = 0;
  rseed(seed);
  TH_PARAM p;
  p.obj = k;
  p.threadId = 0;
  p.isRunning = true;
  p.hasStarted = false;
  p.isWaiting = false;
  p.nbKangaroo = grp;
  p.px.resize(grp);
  p.py.resize(grp);
  p.distance.resize(grp);
  k->CreateHerd(grp, p.px.data(), p.py.data(), p.distance.data(), TAME);
  fprintf(f, "{\"name\":\"%s\",\"rangePower\":%d,\"rangeStart\":\"%s\",\"rangeEnd\":\"%s\",\"priv\":\"%s\","
             "\"pubx\":\"%s\",\"puby\":\"%s\",\"dp\":%d,\"dpMask\":\"%016" PRIx64 "\",\"grp\":%d,\"seed\":%u,\n",
          name, k->rangePower, hx(start).c_str(), hx(end).c_str(), hx(priv).c_str(), hx(pub.x).c_str(),
          hx(pub.y).c_str(), dp, k->dMask, grp, seed);
  fprintf(f, "\"keyx\":\"%s\",\"keyy\":\"%s\",\n", hx(k->keyToSearch.x).c_str(), hx(k->keyToSearch.y).c_str());
  fprintf(f, "\"herd\":[");
  for (int i = 0; i < grp; i++)
    fprintf(f, "%s[\"%s\",\"%s\",\"%s\"]", i ? "," : "", hx(p.px[i]).c_str(), hx(p.py[i]).c_str(),
            hx(p.distance[i]).c_str());
  fprintf(f, "],\n");
  k->endOfSearch = false;
  k->SolveKeyCPU(&p);  // the reference's own CPU walk
  fprintf(f, "\"solved\":%s,\"jumps\":%" PRIu64 ",\"dead\":%" PRIu64 ",\n", k->endOfSearch ? "true" : "false",
          k->counters[0], k->collisionInSameHerd);
  // dump the DP table the walk built (memory/hashtable.cpp layout: x[0..3], d[0..3] packed)
  fprintf(f, "\"table\":[");
  bool first = true;
  uint64_t cnt = 0;
  for (uint32_t h = 0; h < HASH_SIZE; h++) {
    for (uint32_t i = 0; i < k->hashTable.E[h].nbItem; i++) {
      const ENTRY *e = k->hashTable.GetEntry(h, i);
      fprintf(f, "%s[%u,\"%s\",\"%s\"]", first ? "" : ",", h, hx4(e->x.i64).c_str(), hx4(e->d.i64).c_str());
      first = false;
      cnt++;
    }
  }
  fprintf(f, "],\"tableCount\":%" PRIu64 "}%s\n", cnt, last ? "" : ",");
  delete k;
}

// Plain multi-step walk using the reference's Int/IntGroup calls in the order of
// core/kangaroo.cpp:288-325, recording every state — pins per-step (x,y,d) values.
static void stepFixture(FILE *f, Ctx &c) {
  Kangaroo *k = makeKangaroo(c, 4);
  k->rangeStart.SetInt32(0);
  k->rangeEnd.SetBase16("FFFFFFFFFFFFFFFF");
  k->InitRange();
  k->CreateJumpTable();
  Int d0;
  d0.SetBase16("0123456789ABCDEF");
  Point P = c.secp->ComputePublicKey(&d0);
  Int px(&P.x), py(&P.y), d(&d0);
  fprintf(f, "\"steps\":{\"rangePower\":64,\"d0\":\"%s\",\"trace\":[", hx(d0).c_str());
  for (int s = 0; s < 12; s++) {
    uint64_t j = px.bits64[0] & (NB_JUMP - 1);
    Int dx, dy, _s, _p, rx, ry;
    dx.ModSub(&px, &k->jumpPointx[j]);
    IntGroup g(1);
    g.Set(&dx);
    g.ModInv();
    dy.ModSub(&py, &k->jumpPointy[j]);
    _s.ModMulK1(&dy, &dx);
    _p.ModSquareK1(&_s);
    rx.ModSub(&_p, &k->jumpPointx[j]);
    rx.ModSub(&px);
    ry.ModSub(&px, &rx);
    ry.ModMulK1(&_s);
    ry.ModSub(&py);
    px.Set(&rx);
    py.Set(&ry);
    d.ModAddK1order(&k->jumpDistance[j]);
    fprintf(f, "%s[%d,\"%s\",\"%s\",\"%s\"]", s ? "," : "", (int)j, hx(px).c_str(), hx(py).c_str(),
            hx(d).c_str());
  }
  fprintf(f, "]},\n");
  delete k;
}

static void dumpTable(FILE *f, Kangaroo *k) {
  fprintf(f, "\"table\":[");
  bool first = true;
  for (uint32_t h = 0; h < HASH_SIZE; h++)
    for (uint32_t i = 0; i < k->hashTable.E[h].nbItem; i++) {
      const ENTRY *e = k->hashTable.GetEntry(h, i);
      fprintf(f, "%s[%u,\"%s\",\"%s\"]", first ? "" : ",", h, hx4(e->x.i64).c_str(), hx4(e->d.i64).c_str());
      first = false;
    }
  fprintf(f, "]");
}

// A work file written by the reference's own code (storage/backup.cpp:256-428, memory/hashtable.cpp:262-284): a real
// SolveKeyCPU run on a range that does not start at 0 leaves its DP table behind, a fresh herd stands in for the
// kangaroos of a parked thread, and Kangaroo::SaveWork(totalCount, totalTime, threads, nbThread) writes header, table
// and kangaroos.  The JSON beside it holds the same state as the reference has it in memory.
static int saveWorkFixture(Ctx &c, const char *binPath, const char *jsonPath) {
  const int dp = 8, grp = 64, bits = 32;
  std::string wf = binPath, none;
  Kangaroo *k = new Kangaroo(c.secp, dp, false, wf, none, 60, true, false, 0.0, 3000, 17403, 3000, "", "/dev/null", false);
  Int start, end, priv, w;
  start.SetBase16((char *)"5000000000");
  end.Set(&start);
  w.SetInt32(1);
  w.ShiftL(bits);
  w.SubOne();
  end.Add(&w);
  priv.SetBase16((char *)"50B7E15163");
  Point pub = c.secp->ComputePublicKey(&priv);
  if (!k->SetSearchContext(start, end, pub, "savework")) return 2;
  k->CreateJumpTable();
  k->SetDP(dp);
  k->CPU_GRP_SIZE = grp;
  k->counters.fill(0);
  k->nbCPUThread = 1;
  k->nbGPUThread = 0;
  rseed(2024);
  TH_PARAM p;
  p.obj = k;
  p.threadId = 0;
  p.isRunning = true;
  p.hasStarted = false;
  p.isWaiting = false;
  p.nbKangaroo = grp;
  k->endOfSearch = false;
  k->SolveKeyCPU(&p);  // creates its own herd, builds the table, clears the herd when it ends
  p.px.resize(grp);
  p.py.resize(grp);
  p.distance.resize(grp);
  k->CreateHerd(grp, p.px.data(), p.py.data(), p.distance.data(), TAME);
  p.isWaiting = true;  // the state SaveWork waits for (core/thread.cpp:36-42)
  const uint64_t totalCount = k->counters[0];
  const double totalTime = 12.625;
  k->SaveWork(totalCount, totalTime, &p, 1);
  FILE *f = fopen(jsonPath, "w");
  if (!f) return 1;
  fprintf(f, "{\"dp\":%d,\"rangeStart\":\"%s\",\"rangeEnd\":\"%s\",\"pubx\":\"%s\",\"puby\":\"%s\",\"priv\":\"%s\","
             "\"totalCount\":%" PRIu64 ",\"totalTime\":%.6f,\"solved\":%s,\n",
          dp, hx(start).c_str(), hx(end).c_str(), hx(pub.x).c_str(), hx(pub.y).c_str(), hx(priv).c_str(), totalCount,
          totalTime, k->endOfSearch ? "true" : "false");
  dumpTable(f, k);
  fprintf(f, ",\n\"herd\":[");
  for (int i = 0; i < grp; i++)
    fprintf(f, "%s[\"%s\",\"%s\",\"%s\"]", i ? "," : "", hx(p.px[i]).c_str(), hx(p.py[i]).c_str(), hx(p.distance[i]).c_str());
  fprintf(f, "]}\n");
  fclose(f);
  delete k;
  return 0;
}

// The reference's own LoadWork + FetchWalks (storage/backup.cpp:101-150, memory/hashtable.cpp:306-324) read a work file
// (one this repository's kgh_work_save wrote) and what they got is dumped.
static int loadWorkDump(Ctx &c, const char *binPath, const char *jsonPath) {
  Kangaroo *k = makeKangaroo(c, -1);
  std::string fn = binPath;
  if (!k->LoadWork(fn)) return 3;
  const uint64_t n = k->nbLoadedWalk;
  std::vector<Int> x(n ? n : 1), y(n ? n : 1), d(n ? n : 1);
  if (n) k->FetchWalks(n, x.data(), y.data(), d.data());
  FILE *f = fopen(jsonPath, "w");
  if (!f) return 1;
  fprintf(f, "{\"dp\":%d,\"rangeStart\":\"%s\",\"rangeEnd\":\"%s\",\"pubx\":\"%s\",\"puby\":\"%s\","
             "\"totalCount\":%" PRIu64 ",\"totalTime\":%.6f,\"left\":%" PRIu64 ",\n",
          k->initDPSize, hx(k->rangeStart).c_str(), hx(k->rangeEnd).c_str(), hx(k->keysToSearch[0].x).c_str(),
          hx(k->keysToSearch[0].y).c_str(), k->offsetCount, k->offsetTime, k->nbLoadedWalk);
  dumpTable(f, k);
  fprintf(f, ",\n\"herd\":[");
  for (uint64_t i = 0; i < n; i++)
    fprintf(f, "%s[\"%s\",\"%s\",\"%s\"]", i ? "," : "", hx(x[i]).c_str(), hx(y[i]).c_str(), hx(d[i]).c_str());
  fprintf(f, "]}\n");
  fclose(f);
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s out.json | --savework out.bin out.json | --loadwork in.bin out.json\n", argv[0]);
    return 1;
  }
  Timer::Init();
  Secp256K1 *secp = new Secp256K1();
  secp->Init();
  Ctx c{secp, ""};
  if (argc == 4 && !strcmp(argv[1], "--savework")) return saveWorkFixture(c, argv[2], argv[3]);
  if (argc == 4 && !strcmp(argv[1], "--loadwork")) return loadWorkDump(c, argv[2], argv[3]);
  FILE *f = fopen(argv[1], "w");
  if (!f) return 1;
  fprintf(f, "{\n");
  fieldKats(f);
  orderKats(f, *secp);
  groupInvKat(f);
  rngKats(f);
  ecKats(f, *secp);
  jumpTables(f, c);
  stepFixture(f, c);
  fprintf(f, "\"solves\":[\n");
  solveFixture(f, c, "r32_dp6_g64", 32, "0", "B7E15163", 6, 64, 777, false);
  solveFixture(f, c, "r36_dp6_g128_offset", 36, "100000000000", "100A5B3C1D2E", 6, 128, 4242, false);
  solveFixture(f, c, "r40_dp11_g1024", 40, "0", "A5B3C1D2E7", 11, 1024, 99, true);
  fprintf(f, "]\n}\n");
  fclose(f);
  return 0;
}
