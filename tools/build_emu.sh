#!/bin/sh
# builds the host-emulation libraries of the kernel sources (debug aid, see tests/emu/celt_emu.cpp)
set -e
cd "$(dirname "$0")/.."
F="-O1 -g -fwrapv -std=c++17 -shared -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
g++ $F -o tests/emu/libcelt_emu.so tests/emu/celt_emu.cpp
g++ $F -DCA_HOST_EMU_TRACE -o tests/emu/libcelt_emu_trace.so tests/emu/celt_emu.cpp
