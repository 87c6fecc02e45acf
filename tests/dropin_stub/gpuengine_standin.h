// TEST-ONLY stand-in for the reference's math/gpu/engine.h (:13-54): the declaration of `class GPUEngine` that
// kangaroo_amd/dropin/GPUEngine_amd.cpp implements, so that the drop-in class can be compiled and RUN on the GPU
// box, where the reference tree is absent.  Member signatures are the interface and therefore the reference's;
// the private part holds only what GPUEngine_amd.cpp touches.  A real integration compiles GPUEngine_amd.cpp
// against the reference's own header (`make -C oracle refcli` checks that it links).
#ifndef KG_TEST_ENGINE_H
#define KG_TEST_ENGINE_H
#include <string>
#include <vector>

#include "constants.h"
#include "int.h"

class GPUEngine {
 public:
  // lifetime (engine.cu:67-181)
  GPUEngine(int nbThreadGroup, int nbThreadPerGroup, int gpuId, uint32_t maxFound);
  ~GPUEngine();
  // queries (engine.cu:183-189, 277-279)
  int GetMemory();
  int GetNbThread();
  int GetGroupSize();
  std::string deviceName;
  // configuration and herd (engine.cu:63-65, 281-472)
  void SetWildOffset(Int *offset);
  void SetParams(uint64_t dpMask, Int *distance, Int *px, Int *py);
  void SetKangaroos(Int *px, Int *py, Int *d);
  void GetKangaroos(Int *px, Int *py, Int *d);
  void SetKangaroo(uint64_t kIdx, Int *px, Int *py, Int *d);
  // launches (engine.cu:427-440, 474-539)
  bool callKernel();
  bool callKernelAndWait();
  bool Launch(std::vector<ITEM> &hashFound, bool spinWait = false);
  // statics (engine.cu:191-275)
  static bool GetGridSize(int gpuId, int *x, int *y);
  static void *AllocatePinnedMemory(size_t size);
  static void FreePinnedMemory(void *buff);
  static void PrintCudaInfo();

 private:
  Int wildOffset;
  void *resultReady;  // the kg_engine handle
  uint64_t *inputKangaroo, *inputKangarooPinned, *jumpPinned;
  uint32_t *outputItem, *outputItemPinned;
  uint64_t dpMask;
  uint32_t maxFound, outputSize, kangarooSize, kangarooSizePinned, jumpSize;
  int nbThread, nbThreadPerGroup;
  bool initialised, lostWarning;
};
#endif
