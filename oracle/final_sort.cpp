// final_sort.cpp -- TEST INFRASTRUCTURE ONLY (part of libsd_oracle.so, see sd_oracle.c's header).
// Annotation::finalResult (pipeline/src/speakerDiarizer.cpp:962-978) orders the turns with
//     std::sort(results.begin(), results.end(), [](const Result& s1, const Result& s2){ return s1.start < s2.start; });
// std::sort is not stable: which of two turns with EQUAL start comes first is decided by libstdc++'s
// introsort (insertion sort up to 16 elements, median-of-3 quicksort above).  The oracle therefore calls
// the real std::sort -- same comparator, same element order on entry, same libstdc++ -- instead of
// restating it, so tests can compare the order of equal-start turns exactly.
#include <algorithm>
#include <vector>

struct Result { double start, end; int label; };          // sd.cpp:866-876

extern "C" void orc_final_sort(Result* t, long n)
{
    std::vector<Result> results(t, t + n);
    std::sort(results.begin(), results.end(), [](const Result& s1, const Result& s2) { return s1.start < s2.start; });
    std::copy(results.begin(), results.end(), t);
}
