// shim_check.cpp -- exercises the pml_t / ms_t mirror exactly like the reference's callers do
// (one matching_statistics call per read) and prints the vectors; the GPU tests compare the
// output with the CPU checker's.   usage: shim_check <index prefix> <reads: one per line> <P|M> <doc 0|1>
#include <fstream>
#include <iostream>

#include "spumoni_index.hpp"

using namespace spumoni_host;

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const std::string prefix = argv[1];
    const bool ms = argv[3][0] == 'M', doc = argv[4][0] == '1';
    std::ifstream in(argv[2]);
    std::string read;
    std::vector<size_t> lengths, pointers, docs;
    auto dump = [](const char* tag, const std::vector<size_t>& v) {
        std::cout << tag;
        for (size_t x : v) std::cout << ' ' << x;
        std::cout << '\n';
    };
    if (!ms) {
        pml_t idx(prefix, doc, true);
        auto st = idx.get_bwt_stats();
        std::cout << "stats " << st.first << ' ' << st.second << '\n';
        while (std::getline(in, read)) {
            if (doc)
                idx.matching_statistics(read.c_str(), read.size(), lengths, docs);
            else
                idx.matching_statistics(read.c_str(), read.size(), lengths);
            dump("L", lengths);
            if (doc) dump("D", docs);
        }
    } else {
        ms_t idx(prefix, doc, true);
        auto st = idx.get_bwt_stats();
        std::cout << "stats " << st.first << ' ' << st.second << '\n';
        while (std::getline(in, read)) {
            if (doc)
                idx.matching_statistics(read.c_str(), read.size(), lengths, pointers, docs);
            else
                idx.matching_statistics(read.c_str(), read.size(), lengths, pointers);
            dump("L", lengths);
            dump("P", pointers);
            if (doc) dump("D", docs);
        }
    }
    return 0;
}
