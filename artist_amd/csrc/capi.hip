// capi.hip - the non-kernel part of the C ABI (include/artist_hip.h).
#include "launch_common.hpp"

namespace art {
thread_local int g_last_hip_error = 0;
}

extern "C" int art_abi_version(void) { return 13; }

extern "C" int art_last_hip_error(void) { return art::g_last_hip_error; }

extern "C" const char* art_strerror(int code)
{
    switch (code) {
        case ART_OK: return "ok";
        case ART_EINVAL: return "invalid argument (null pointer, bad size, or unsupported degree)";
        case ART_ETARGET: return "target index out of range";
        case ART_ELAUNCH: return "HIP runtime/launch error (see art_last_hip_error)";
        case ART_EUNSUPPORTED: return "unsupported configuration";
        case ART_ECANDIDATES: return "a heliostat has more blocking rectangles inside its ray cone than the kernels hold";
        case ART_EQUEUE: return "a work counter of this stream was not zero when a trace call started: an earlier launch on the stream ended "
                                "abnormally (the counters were reset; results of that earlier call are not valid)";
        default: return "unknown error";
    }
}
