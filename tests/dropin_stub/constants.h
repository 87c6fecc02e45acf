// TEST-ONLY stand-in for the reference's constants.h (constants.h:12,15,21): just the three values that are part
// of the GPUEngine ABI.  Used by tests/dropin_host.cpp to compile kangaroo_amd/dropin/GPUEngine_amd.cpp on a
// machine where the reference tree is absent (the GPU box).  In a real integration the reference's own header is used.
#ifndef KG_TEST_CONSTANTS_H
#define KG_TEST_CONSTANTS_H
#define NB_JUMP 64
#define GPU_GRP_SIZE 128
#define NB_RUN 256
#endif
