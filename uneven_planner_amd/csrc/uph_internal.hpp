// Internal glue between the two translation units of libunevenhip.so (unevenhip.hip, map_build.hip).
#pragma once
#include <string>

#include "../../include/uneven_hip.h"
#include "uph_common.hpp"

namespace uph {
void setError(const std::string& s);
}

int uphMapDevice(const uph_map* m);
uph::GridDev uphMapGrid(const uph_map* m);

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
