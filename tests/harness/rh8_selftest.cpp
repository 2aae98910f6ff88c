// TEST INFRASTRUCTURE: stdin protocol of oracle/rh_probe.cpp ("u8 k k k ...") answered by the product's
// hs_rh8.h emulator, so that it can be checked against tests/golden/robin_hood_order.json.
#include <iostream>
#include <sstream>
#include <string>
#include "../../hairsplitter_amd/csrc/hs_rh8.h"
int main() {
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
