// Oracle self-test helper: same stdin protocol as rh_probe.cpp but answers with oracle/hs_oracle_rh.h.
#include <iostream>
#include <sstream>
#include <string>
#include "hs_oracle_rh.h"
int main() {
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream iss(line);
        std::string type; iss >> type;
        long k; bool first = true;
        if (type == "u8") {
            hso::RHMap<unsigned char, int> m;
            while (iss >> k) { unsigned char c = (unsigned char)k; if (!m.contains(c)) m[c] = 0; m[c] += 1; }
            m.for_each([&](unsigned char key, int) { std::cout << (first ? "" : " ") << (int)key; first = false; });
        } else {
            hso::RHMap<int, int> m;
            while (iss >> k) { m[(int)k] += 1; }
            m.for_each([&](int key, int) { std::cout << (first ? "" : " ") << key; first = false; });
        }
        std::cout << "\n";
    }
}
