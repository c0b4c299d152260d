// tests/sim/sim_cli.cpp -- TEST INFRASTRUCTURE: the simulator build (sim_main.cpp) as a program, so that it can run under
// AddressSanitizer / UndefinedBehaviorSanitizer (tests/test_sim_cpu.py::test_sim_row_batches_clean_under_asan_ubsan).
// usage: sim_cli <chunk file> <frame file>
#include "sim_main.cpp"
#include <fstream>
#include <iterator>
int main(int argc, char** argv) {
    if (argc != 3) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    std::vector<uint8_t> src((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    std::vector<uint8_t> dst(src.size() + (src.size() >> 8) + 1024);
    size_t const n = sim_compress_chunk(src.data(), src.size(), dst.data(), dst.size(), 1);
    std::ofstream out(argv[2], std::ios::binary);
    out.write((const char*)dst.data(), (std::streamsize)n);
    return n ? 0 : 1;
}
