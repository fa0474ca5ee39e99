// tests/stubhip/hip/hiprtc.h — hipRTC stand-in: "compiles" any source whose text contains the kernel's name, fails otherwise (with a log)
#pragma once
#include <cstddef>
typedef int hiprtcResult;
enum { HIPRTC_SUCCESS = 0, HIPRTC_ERROR_COMPILATION = 6 };
typedef struct stubhip_program *hiprtcProgram;
hiprtcResult hiprtcCreateProgram(hiprtcProgram *p, const char *src, const char *name, int nh, const char **headers, const char **names);
hiprtcResult hiprtcCompileProgram(hiprtcProgram p, int nopts, const char **opts);
hiprtcResult hiprtcGetProgramLogSize(hiprtcProgram p, size_t *n);
hiprtcResult hiprtcGetProgramLog(hiprtcProgram p, char *log);
hiprtcResult hiprtcGetCodeSize(hiprtcProgram p, size_t *n);
hiprtcResult hiprtcGetCode(hiprtcProgram p, char *code);
hiprtcResult hiprtcDestroyProgram(hiprtcProgram *p);
const char *hiprtcGetErrorString(hiprtcResult r);
