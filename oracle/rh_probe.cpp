// Test-infrastructure probe (build-owned): instantiates the REFERENCE's vendored robin_hood.h
// (/root/reference/src/robin_hood.h, included where it lies at build time, never copied) and prints the
// iteration order for key sequences read from stdin. Used by gen_goldens.py to produce
// tests/golden/robin_hood_order.json, which pins oracle/hs_oracle_rh.h and the product's emulator.
// stdin lines: "u8 k k k ..." or "int k k k ..."; stdout: iteration order, one line per input line.
#include <iostream>
#include <sstream>
#include <string>
#include "robin_hood.h"
int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream iss(line);
        std::string type; iss >> type;
        long k;
        if (type == "u8") {
            robin_hood::unordered_map<unsigned char, int> m;
            while (iss >> k) { unsigned char c = (unsigned char)k; if (m.find(c) == m.end()) m[c] = 0; m[c] += 1; }
            bool first = true;
            for (auto& kv : m) { std::cout << (first ? "" : " ") << (int)kv.first; first = false; }
        } else {
            robin_hood::unordered_map<int, int> m;
            while (iss >> k) { m[(int)k] += 1; }
            bool first = true;
            for (auto& kv : m) { std::cout << (first ? "" : " ") << kv.first; first = false; }
        }
        std::cout << "\n";
    }
    return 0;
}
