// libunevenhip.so -- the `.map` cache of UnevenMap in the host's own language: the CSV the reference writes at the end of
// constructMap (uneven_map/src/uneven_map.cpp:400-412) and reads back in constructMapInput (:270-315), plus a binary side-car that holds the
// cells bit for bit (the CSV keeps six significant digits).  Plain host code: no device is touched here; uph_map_save_cache / uph_map_load_cache
// (map_build.hip) move the cells between these files and a device map.
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <locale.h>
#include <string>
#include <vector>

#include "../../include/uneven_hip.h"

namespace uph { void setError(const std::string& s); }      // unevenhip.hip (thread-local text behind uph_last_error)
using uph::setError;

static const char BIN_MAGIC[8] = {'U', 'P', 'H', 'M', 'A', 'P', '0', '1'};

// The numbers of a .map file are written and parsed in the "C" locale whatever the host process has selected with setlocale(): a decimal comma would
// break the format on the way out and truncate every value at the point on the way in.  (The reference's ostream writes with the C++ global locale --
// classic unless the host changes that too -- and parses with stold, i.e. with the C locale in force.)  Per thread (uselocale), restored on scope exit.
struct CNumericLocale {
    locale_t c, old;
    CNumericLocale() : c(newlocale(LC_NUMERIC_MASK, "C", (locale_t)0)), old(c ? uselocale(c) : (locale_t)0) {}
    ~CNumericLocale() { if (c) { uselocale(old); freelocale(c); } }
};

static bool dimsOk(const int32_t d[3]) { return d && d[0] > 0 && d[1] > 0 && d[2] > 0 && (int64_t)d[0] * d[1] * d[2] < ((int64_t)1 << 40); }

extern "C" {

// uneven_map.cpp:400-412: `outf << x << "," << y << "," << yaw << "," << rs2.z << "," << rs2.sigma << "," << rs2.zb.x() << "," << rs2.zb.y() << endl`
// for x, y, yaw ascending (= the address order, uneven_map.h:427-435).  A default-constructed ostream prints a double like printf's %g with
// precision 6, and std::endl ends the line with '\n'.
int uph_map_save_csv(const char* path, const double* rxs2, const int32_t dims3[3]) {
    if (!path || !rxs2 || !dimsOk(dims3)) { setError("uph_map_save_csv: bad arguments"); return UPH_ERR_INVALID; }
    FILE* f = std::fopen(path, "w");
    if (!f) { setError(std::string("uph_map_save_csv: cannot open ") + path + ": " + std::strerror(errno)); return UPH_ERR_INVALID; }
    std::vector<char> buf(1 << 20);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    const CNumericLocale in_c;
    const double* c = rxs2;
    bool ok = true;
    for (int x = 0; x < dims3[0] && ok; x++)
        for (int y = 0; y < dims3[1] && ok; y++)
            for (int w = 0; w < dims3[2]; w++, c += 4)
                if (std::fprintf(f, "%d,%d,%d,%g,%g,%g,%g\n", x, y, w, c[0], c[1], c[2], c[3]) < 0) { ok = false; break; }
    if (std::fclose(f) != 0) ok = false;
    if (!ok) { setError(std::string("uph_map_save_csv: write to ") + path + " failed: " + std::strerror(errno)); return UPH_ERR_INVALID; }
    return UPH_OK;
}

// uneven_map.cpp:270-315.  The cells the file does not mention stay RXS2() = zeros with c = 1 (uneven_map.cpp:117-119); a line is split at the
// commas, the three indices go through atoi, the four values through stold and are then narrowed to double (TWO roundings: a direct
// string -> double conversion differs by an ulp on ~1e-4 of the values), an index outside the grid drops the line (isInMap(Vector3i), :300);
// later lines overwrite earlier ones.  (`if (map_buffer[..].sigma < sigma) sigma = sigma`, :303-306, changes nothing.)  A line with fewer
// than seven fields -- undefined behaviour in the reference -- is skipped.  Returns UPH_ERR_INVALID when the file cannot be opened: the
// reference then builds the map (`if (!constructMapInput()) constructMap()`, :166-167).
int uph_map_load_csv(const char* path, const int32_t dims3[3], double* rxs2, double* c, int64_t* n_lines) {
    if (!path || !rxs2 || !dimsOk(dims3)) { setError("uph_map_load_csv: bad arguments"); return UPH_ERR_INVALID; }
    FILE* f = std::fopen(path, "r");
    if (!f) { setError(std::string("uph_map_load_csv: cannot open ") + path + ": " + std::strerror(errno)); return UPH_ERR_INVALID; }
    std::vector<char> buf(1 << 20);
    std::setvbuf(f, buf.data(), _IOFBF, buf.size());
    const int64_t ncell = (int64_t)dims3[0] * dims3[1] * dims3[2];
    std::memset(rxs2, 0, sizeof(double) * 4 * (size_t)ncell);
    if (c) for (int64_t i = 0; i < ncell; i++) c[i] = 1.0;
    int64_t used = 0;
    const CNumericLocale in_c;
    char* line = nullptr;
    size_t cap = 0;
    ssize_t len;
    while ((len = getline(&line, &cap, f)) >= 0) {
        char* w[7];
        int nw = 0;
        char* p = line;
        // getline(sin, word, ','): fields end at a comma; the last one runs to the end of the line
        while (nw < 7) {
            w[nw++] = p;
            char* q = std::strchr(p, ',');
            if (!q) break;
            *q = 0;
            p = q + 1;
        }
        if (nw < 7) continue;
        const int x = std::atoi(w[0]), y = std::atoi(w[1]), yw = std::atoi(w[2]);
        const double z = (double)std::strtold(w[3], nullptr), sg = (double)std::strtold(w[4], nullptr);
        const double za = (double)std::strtold(w[5], nullptr), zb = (double)std::strtold(w[6], nullptr);
        if (x < 0 || y < 0 || yw < 0 || x >= dims3[0] || y >= dims3[1] || yw >= dims3[2]) continue;
        const size_t a = ((size_t)x * dims3[1] + y) * dims3[2] + yw;
        rxs2[4 * a] = z; rxs2[4 * a + 1] = sg; rxs2[4 * a + 2] = za; rxs2[4 * a + 3] = zb;
        if (c) c[a] = std::sqrt(1.0 - za * za - zb * zb);
        used++;
    }
    std::free(line);
    std::fclose(f);
    if (n_lines) *n_lines = used;
    return UPH_OK;
}

// binary side-car: "UPHMAP01", the three dimensions as little-endian int64, then ncell x 4 doubles in address order -- the built grid bit for bit
int uph_map_save_bin(const char* path, const double* rxs2, const int32_t dims3[3]) {
    if (!path || !rxs2 || !dimsOk(dims3)) { setError("uph_map_save_bin: bad arguments"); return UPH_ERR_INVALID; }
    FILE* f = std::fopen(path, "wb");
    if (!f) { setError(std::string("uph_map_save_bin: cannot open ") + path + ": " + std::strerror(errno)); return UPH_ERR_INVALID; }
    const int64_t d[3] = {dims3[0], dims3[1], dims3[2]};
    const size_t n = (size_t)d[0] * d[1] * d[2] * 4;
    bool ok = std::fwrite(BIN_MAGIC, 1, 8, f) == 8 && std::fwrite(d, 8, 3, f) == 3 && std::fwrite(rxs2, 8, n, f) == n;
    if (std::fclose(f) != 0) ok = false;
    if (!ok) { setError(std::string("uph_map_save_bin: write to ") + path + " failed"); return UPH_ERR_INVALID; }
    return UPH_OK;
}

// UPH_ERR_INVALID: cannot be opened / not a side-car;  UPH_ERR_LIMIT: written for a grid of other dimensions, or truncated
int uph_map_load_bin(const char* path, const int32_t dims3[3], double* rxs2) {
    if (!path || !rxs2 || !dimsOk(dims3)) { setError("uph_map_load_bin: bad arguments"); return UPH_ERR_INVALID; }
    FILE* f = std::fopen(path, "rb");
    if (!f) { setError(std::string("uph_map_load_bin: cannot open ") + path + ": " + std::strerror(errno)); return UPH_ERR_INVALID; }
    char magic[8];
    int64_t d[3];
    int rc = UPH_OK;
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, BIN_MAGIC, 8) != 0 || std::fread(d, 8, 3, f) != 3) {
        setError(std::string("uph_map_load_bin: ") + path + " is not a binary .map side-car");
        rc = UPH_ERR_INVALID;
    } else if (d[0] != dims3[0] || d[1] != dims3[1] || d[2] != dims3[2]) {
        setError(std::string("uph_map_load_bin: ") + path + " was written for a " + std::to_string(d[0]) + " x " + std::to_string(d[1]) + " x " + std::to_string(d[2]) + " grid");
        rc = UPH_ERR_LIMIT;
    } else {
        const size_t n = (size_t)d[0] * d[1] * d[2] * 4;
        if (std::fread(rxs2, 8, n, f) != n) { setError(std::string("uph_map_load_bin: ") + path + " is truncated"); rc = UPH_ERR_LIMIT; }
    }
    std::fclose(f);
    return rc;
}

}  // extern "C"
