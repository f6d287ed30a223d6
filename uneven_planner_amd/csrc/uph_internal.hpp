// Internal glue between the two translation units of libunevenhip.so (unevenhip.hip, map_build.hip).
#pragma once
#include <string>

#include "../../include/uneven_hip.h"
#include "uph_common.hpp"

namespace uph {
void setError(const std::string& s);
}

int uphMapDevice(const uph_map* m);
// grow-only device scratch owned by the map (slots 0..3), for the batched query entry points: no allocation per call once warm.
// One query at a time per map.  nullptr (with uph_last_error set) when the allocation fails.
void* uphMapScratch(uph_map* m, int slot, size_t bytes);
struct UphPtr {                 // non-owning view of a scratch slot
    void* p = nullptr;
    template <class T> T* as() { return (T*)p; }
};
uph::GridDev uphMapGrid(const uph_map* m);
// device occupancy layers of the map (uneven_map.cpp:170-179): occ [ncell], occ_r2 [nx * ny]; read by the front-end search (kino_search.hip)
void uphMapOcc(const uph_map* m, const char** occ, const char** occ_r2);

// scope guards for the temporaries of the extern "C" entry points: every early return (HIPCHK) releases them
struct UphDevTmp {
    void* p = nullptr;
    ~UphDevTmp();
    template <class T> T* as() { return (T*)p; }
};
struct UphEventTmp {
    void* e = nullptr;          // hipEvent_t
    ~UphEventTmp();
};
