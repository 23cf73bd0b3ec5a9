// Per-device state of the library, keyed by the device ordinal: one slot per GPU a process may drive (one Engine per device,
// one host thread per device: include/endosurf_hip.h).  Plain C++ (no HIP) so that the table logic is unit-tested on the CPU
// (tests/test_host_logic.py compiles it with g++).
#pragma once

namespace es {

template <class T, int MAX_DEVICES = 64>
struct DeviceTable {
    T slot[MAX_DEVICES];
    // out-of-range ordinals share the last slot (never out of bounds; a process with > 64 GPUs does not exist on this platform)
    T& at(int dev) { return slot[dev < 0 ? 0 : (dev < MAX_DEVICES ? dev : MAX_DEVICES - 1)]; }
};

}  // namespace es
