// TEST INFRASTRUCTURE: stdin protocol of oracle/rh_probe.cpp ("u8 k k k ...") answered by the product's
// hs_rh8.h emulator, so that it can be checked against tests/golden/robin_hood_order.json.
#include <iostream>
#include <sstream>
#include <string>
#include <cstring>
#include <algorithm>
#include "../../hairsplitter_amd/csrc/hs_rh8.h"
int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "worstcase") {
        // the device keeps the map in 512 bytes of LDS per table (hs::Rh8View, cap 512): every set of byte keys the path can produce -- up to 125
        // pileup codes + the three fillers / the reference code -- must fit without the overflow flag and iterate as the unbounded map does
        unsigned long long x = 88172645463325252ull;
        auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
        static uint8_t info[512], key[512], tmp[512];
        for (int trial = 0; trial < 20000; ++trial) {
            const int n = trial < 2000 ? 128 : 1 + (int)(rnd() % 128);
            uint8_t keys[256]; for (int i = 0; i < 256; ++i) keys[i] = (uint8_t)i;
            for (int i = 255; i > 0; --i) { const int j = (int)(rnd() % (unsigned)(i + 1)); std::swap(keys[i], keys[j]); }
            hs::Rh8 a; a.clear();
            hs::Rh8View v; v.init(info, key, tmp, 512);
            for (int i = 0; i < n; ++i) { a.insert(keys[i]); v.insert(keys[i]); }
            if (v.overflow) { std::cout << "overflow at " << n << " keys\n"; return 1; }
            uint8_t oa[300], ov[512];
            const int na = a.order(oa), nv = v.order(ov);
            if (na != nv || std::memcmp(oa, ov, (size_t)na) != 0) { std::cout << "order differs at " << n << " keys\n"; return 1; }
        }
        std::cout << "ok\n";
        return 0;
    }
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream iss(line);
        std::string type; iss >> type;
        if (type != "u8") { std::cout << "\n"; continue; }
        hs::Rh8 rh; rh.clear();
        long k;
        while (iss >> k) rh.insert((uint8_t)k);
        uint8_t ord[300];
        int n = rh.order(ord);
        for (int i = 0; i < n; ++i) std::cout << (i ? " " : "") << (int)ord[i];
        std::cout << "\n";
    }
}
