// api.hip -- library identification and error text for lib3pu_hip.so.
#include "tpu3_dev.h"

extern "C" const char *tpu3_version(void)
{
    return "3pu-hip 0.1.0 gfx950";
}

extern "C" const char *tpu3_strerror(int code)
{
    if (code == TPU3_OK) return "ok";
    if (code == TPU3_EINVAL) return "invalid argument (size, NULL pointer or element size)";
    if (code == TPU3_ELIMIT) return "size beyond what this build supports";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown tpu3 error";
}
