// build_info.hip — the identity of this build of libsgam_hip.so, stamped by sgam_neurips22_amd/build.py at compile time:
//   SGAM_BUILD_COMMIT  the last commit that touched the library's sources (csrc/, include/, build.py), "+dirty" when the
//                      working tree differed from it; taken from git where the build runs (the GPU box has no .git: the
//                      library travels prebuilt and carries the stamp with it)
//   SGAM_BUILD_DIGEST  sha256 (first 12 hex digits) over every source, header and compile flag of the library
// bench.py prints them as `head` / `lib_digest` and compares them with the stamp of the committed counter files.
#include "sgam_common.h"

#ifndef SGAM_BUILD_COMMIT
#define SGAM_BUILD_COMMIT "unknown"
#endif
#ifndef SGAM_BUILD_DIGEST
#define SGAM_BUILD_DIGEST "unknown"
#endif

extern "C" int sgam_abi_version(void) { return 10; }
extern "C" const char *sgam_build_info(void) {
    return "libsgam_hip gfx950 (CDNA4): split-fp32 / fp32-in / 16-bit MFMA paths, built " __DATE__
           "; commit " SGAM_BUILD_COMMIT "; digest " SGAM_BUILD_DIGEST;
}
extern "C" const char *sgam_build_commit(void) { return SGAM_BUILD_COMMIT; }
extern "C" const char *sgam_build_digest(void) { return SGAM_BUILD_DIGEST; }
