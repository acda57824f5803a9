// device_tables.h -- the generated CELT tables (celt_tables.h) as device-global constants
// (plain static const arrays in the CA_HOST_EMU build). Arrays are tiny and L2-resident.
#pragma once
#include "wave.h"
#define CLT_TABLE_QUAL CA_DEVICE_CONST
#include "celt_tables.h"
