// opusgpu_internal.h -- helpers shared by the translation units of libopusgpu.so (not exported API).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/opusgpu.h"

extern "C" void opusgpu_set_last_error(int err);
// hipGetLastError() mapped to an OPUSGPU_* code
extern "C" int opusgpu_check_launch(void);
