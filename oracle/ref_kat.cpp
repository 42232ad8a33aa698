// oracle/ref_kat.cpp -- TEST INFRASTRUCTURE, build container only.
// Known-answer driver around the REFERENCE's own headers (compiled in place
// from /root/reference/src by oracle/Makefile into oracle/_ref/ref_kat).
// It exists to mint golden vectors for rows 1-5 of SURVEY.md section 8a; it is
// never built or run on the GPU box and contains no reference source.
//
// stdin protocol, one command per line; one output line per command:
//   H x            -> thomas_mueller_hash(x)                    (src/hash_int.h:39)
//   R x k          -> make_reverse_complement(uint32 x, k)      (src/dna_encoding.h:113)
//   C x k          -> make_canonical(uint32 x, k)               (src/dna_encoding.h:187)
//   W n len stride -> "b:e b:e ..." windows of for_each_window  (src/dna_encoding.h:259)
//   S k s seq      -> sketch values of the sketcher             (src/hash_dna.h:113)
//                     (seq "-" = empty string)
#include <cstdint>
#include <iostream>
#include <sstream>
#include <string>

#include "hash_int.h"
#include "dna_encoding.h"
#include "hash_dna.h"

// unqualified call: the overloads live in an unnamed namespace inside mc, which a
// qualified mc:: lookup does not reach (it stops at mc's deleted catch-all).
namespace mc { inline std::uint32_t ref_revcomp(std::uint32_t x, numk_t k) { return make_reverse_complement(x, k); } }

int main() {
    std::ios::sync_with_stdio(false);
    std::string line;
    while (std::getline(std::cin, line)) {
        if (line.empty()) continue;
        std::istringstream is(line);
        char cmd; is >> cmd;
        if (cmd == 'H') {
            std::uint32_t x; is >> x;
            std::cout << mc::thomas_mueller_hash(x) << '\n';
        } else if (cmd == 'R') {
            std::uint32_t x; unsigned k; is >> x >> k;
            std::cout << mc::ref_revcomp(x, mc::numk_t(k)) << '\n';
        } else if (cmd == 'C') {
            std::uint32_t x; unsigned k; is >> x >> k;
            std::cout << mc::make_canonical(x, mc::numk_t(k)) << '\n';
        } else if (cmd == 'W') {
            std::size_t n, len, stride; is >> n >> len >> stride;
            std::string s(n, 'A');
            bool first = true;
            mc::for_each_window(s.begin(), s.end(), len, stride,
                [&](std::string::iterator b, std::string::iterator e) {
                    if (!first) std::cout << ' ';
                    first = false;
                    std::cout << (b - s.begin()) << ':' << (e - s.begin());
                });
            std::cout << '\n';
        } else if (cmd == 'S') {
            unsigned k, s; std::string seq; is >> k >> s >> seq;
            if (seq == "-") seq.clear();
            mc::single_function_unique_min_hasher<std::uint32_t> sk;
            sk.kmer_size(mc::numk_t(k));
            sk.sketch_size(s);
            auto v = sk(seq);
            bool first = true;
            for (auto f : v) { if (!first) std::cout << ' '; first = false; std::cout << f; }
            std::cout << '\n';
        } else {
            std::cout << "?\n";
        }
    }
    return 0;
}
