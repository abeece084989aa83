// gen_reference_inputs.cpp -- regenerates the INPUTS of the reference's seeded unit tests for the
// structures on the k-mer counting path (own code; nothing is taken from the reference but the
// seeds, sizes, thresholds and the order of the random draws, each cited below).
//
// The reference's tests draw from std::mt19937 through std::uniform_real_distribution<> /
// std::uniform_int_distribution<> of libstdc++; this program uses the same library types, so it
// reproduces the same streams wherever g++/libstdc++ is the toolchain (it is in this image and
// on the GPU box).  tests/test_reference_vectors.py compiles and runs it, checks the streams
// against the digests committed in tests/golden/reference_kat.json, and replays the assertions
// of the reference's tests over files written by the oracle and by the product.
//
// Output: one case per line: <name> <kind> <count> v0 v1 ...   (kind: ones = ascending bit
// positions, pos128 = positions as lo:hi hex pairs, values = u32 values)
//
// Build: g++ -O2 -std=c++17 -o gen_reference_inputs gen_reference_inputs.cpp
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

namespace {

// ones of a Bernoulli stream: bit i is set iff pred(dist(rng)); one draw per position
template <class Pred>
void bernoulli(const char* name, uint32_t seed, uint64_t nbits, Pred pred)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<> dist;
    std::vector<uint64_t> ones;
    for (uint64_t i = 0; i < nbits; ++i)
        if (pred(dist(rng))) ones.push_back(i);
    std::printf("%s ones %zu %llu", name, ones.size(), (unsigned long long)nbits);
    for (uint64_t p : ones) std::printf(" %llu", (unsigned long long)p);
    std::printf("\n");
}

void literal(const char* name, uint64_t nbits, std::initializer_list<uint64_t> ones)
{
    std::printf("%s ones %zu %llu", name, ones.size(), (unsigned long long)nbits);
    for (uint64_t p : ones) std::printf(" %llu", (unsigned long long)p);
    std::printf("\n");
}

// testSparseArray.cc test3 (:158-170): position i = (i << 64) | rng() << 32 | rng(); test4/5
// (:205-221, :256-272): that value << 28 | (rng() & (2^28 - 1))
void wide_positions(const char* name, uint32_t seed, uint64_t m, bool shift28)
{
    std::mt19937 rng(seed);
    std::printf("%s pos128 %llu", name, (unsigned long long)m);
    for (uint64_t i = 0; i < m; ++i)
    {
        unsigned __int128 v = i;
        v <<= 64;
        v |= static_cast<uint64_t>(rng()) << 32;
        v |= static_cast<uint64_t>(rng());
        if (shift28)
        {
            v <<= 28;
            v |= static_cast<uint64_t>(rng()) & ((1ULL << 28) - 1);
        }
        std::printf(" %llx:%llx", (unsigned long long)(uint64_t)v, (unsigned long long)(uint64_t)(v >> 64));
    }
    std::printf("\n");
}

}  // namespace

int main()
{
    // ---- testSparseArray.cc ---------------------------------------------------------------
    bernoulli("sparse_test1", 17, 30, [](double x) { return x < 0.1; });              // :41-63  N = 30, M = N * 0.1
    bernoulli("sparse_test2", 17, 1000, [](double x) { return x < 0.01; });           // :117-138
    wide_positions("sparse_test3", 17, 120, false);                                    // :154-172 N = 2^72
    wide_positions("sparse_test4", 17, 120, true);                                     // :202-223 N = 2^100 (test5 :253-274 draws the same)

    // ---- testDenseArray.cc ----------------------------------------------------------------
    bernoulli("dense_test1", 17, 100000, [](double x) { return x > 0.5; });           // :82-97  DenseArray
    bernoulli("dense_test2", 17, 1000000, [](double x) { return x < 1.0 / 70000.0; });// :137-163 also testWordyBitVector.cc test4/5 :132-150,:172-190
    bernoulli("dense_test3", 17, 100000, [](double x) { return x < 0.999; });         // :196-216
    bernoulli("dense_test4", 17, 20, [](double x) { return x > 0.5; });               // :243-262 inverted sense
    bernoulli("dense_test5", 17, 100000, [](double x) { return x > 0.5; });           // :289-308 inverted sense
    literal("dense_test6", 20, {0, 1, 4, 6, 8, 10, 12, 15, 17, 18});                   // :340-358
    bernoulli("dense_one_in_10", 17, 1000000, [](double x) { return x < 0.1; });      // :384-403
    bernoulli("dense_one_in_100", 17, 1000000, [](double x) { return x < 0.01; });    // :443-462
    bernoulli("dense_one_in_1000", 17, 10000000, [](double x) { return x < 0.001; }); // :502-521
    bernoulli("dense_one_in_10000", 17, 10000000, [](double x) { return x < 0.0001; });// :561-580
    literal("dense_bug_over_256", 516, {});                                            // :26-40 every bit set: written out by the test itself

    // ---- testWordyBitVector.cc ------------------------------------------------------------
    literal("wordy_test2", 236, {7, 47, 63, 64, 65, 97, 108, 235});                    // :44-58
    literal("wordy_test3", 18, {2, 3, 5, 7, 8, 10, 12, 14, 15, 17});                   // :104-118

    // ---- testVariableByteArray.cc ---------------------------------------------------------
    {
        const uint32_t v[] = {0, 1, 2, 3, 4, 254, 255, 256, 257, 1, 2, 3, 65535, 65536, 3, 65535};   // test1 :27-47
        std::printf("vba_test1 values 16");
        for (uint32_t x : v) std::printf(" %u", x);
        std::printf("\n");
    }
    for (int t = 0; t < 2; ++t)
    {
        const uint64_t n = t == 0 ? 10000 : 1000;                                       // test2 :66-86, test3 :93-113
        std::mt19937 rng(209);
        std::uniform_int_distribution<> dist(0, 70000);
        std::printf("%s values %llu", t == 0 ? "vba_test2" : "vba_test3", (unsigned long long)n);
        for (uint64_t i = 0; i < n; ++i) std::printf(" %u", (uint32_t)dist(rng));
        std::printf("\n");
    }
    {
        const uint64_t n = 100000;                                                      // test4 :124-148
        std::mt19937 rng(209);
        std::uniform_real_distribution<> dist;
        std::printf("vba_test4 values %llu", (unsigned long long)n);
        for (uint64_t i = 0; i < n; ++i)
        {
            double x = dist(rng);
            uint64_t y = x * x * x * 1024 * 1024 * 16;
            std::printf(" %u", (uint32_t)y);
        }
        std::printf("\n");
    }
    return 0;
}
