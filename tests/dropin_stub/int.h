// TEST-ONLY stand-in for the reference's Int and ITEM (math/cpu/int.h:165-168, 339-343): the storage layout the
// GPUEngine interface exchanges (five 64-bit limbs, little endian, limb 4 = overflow word) and the two setters
// GPUEngine_amd.cpp calls.  No arithmetic: the drop-in class does none on Int.
#ifndef KG_TEST_INT_H
#define KG_TEST_INT_H
#include <stdint.h>
#include <string.h>

class Int {
 public:
  Int() { memset(bits64, 0, sizeof bits64); }
  void SetInt32(uint32_t v) {
    memset(bits64, 0, sizeof bits64);
    bits64[0] = v;
  }
  void Set(Int *o) { memcpy(bits64, o->bits64, sizeof bits64); }
  union {
    uint32_t bits[10];
    uint64_t bits64[5];
  };
};

struct ITEM {
  Int x;
  Int d;
  uint64_t kIdx;
};
#endif
