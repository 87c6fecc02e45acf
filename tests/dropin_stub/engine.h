// TEST-ONLY: the name kangaroo_amd/dropin/GPUEngine_amd.cpp includes.  In a real integration this is the reference's
// own math/gpu/engine.h; on the GPU box, where the reference tree is absent, it forwards to the stand-in declaration.
#include "gpuengine_standin.h"
