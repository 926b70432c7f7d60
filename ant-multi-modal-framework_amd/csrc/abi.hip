// abi.hip -- identification entry points of the C ABI (include/antmmf_hip.h).
#include "common.h"
extern "C" int antmmf_backend(void) {
#ifdef ANTMMF_EMULATE
    return 0;  // CPU lane emulator (tests/emu only)
#else
    return 1;  // gfx950 device code
#endif
}
extern "C" int antmmf_abi_version(void) { return 2; }   // 2: the sub-LN fold entry points moved to the lab library (include/antmmf_hip_lab.h)
