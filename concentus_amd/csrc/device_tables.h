// device_tables.h -- the generated CELT tables (celt_tables.h) as device-global constants.
// One definition per translation unit that includes it; arrays are tiny and L2-resident.
#pragma once
#include <hip/hip_runtime.h>
#define CLT_TABLE_QUAL static __device__ const
#include "celt_tables.h"
