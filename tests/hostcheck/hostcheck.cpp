// Unit-test shim: compiles the sequential device functions of image-analysis_amd/csrc/lfa_algo.h for the HOST so their
// arithmetic can be checked against the golden vectors on a machine without a GPU.  Test infrastructure only: nothing
// in the product loads this library.
#include "../../image-analysis_amd/csrc/lfa_algo.h"

extern "C" {

int hc_two_peaks(const double* p, int n, int inverse, int rng, int fallback, int window,
                 int* start, int* end, double* area) {
    lfa::TwoPeaks t = lfa::two_peaks(p, n, inverse != 0, rng, fallback != 0, window);
    for (int i = 0; i < t.n; ++i) { start[i] = t.pk[i].start; end[i] = t.pk[i].end; area[i] = t.pk[i].area; }
    return t.n;
}

void hc_analyze_peaks(const double* p, int n, int inverse, int window, int* start, int* end, double* area) {
    lfa::Peak k = lfa::analyze_peaks(p, n, inverse != 0, window);
    *start = k.start; *end = k.end; *area = k.area;
}

double hc_pairwise(const double* p, int n) { return lfa::np_pairwise(lfa::ArrAt{p}, 0, n); }

int hc_otsu_float256(const uint32_t* cnt, double vmin, double vmax, double* centre) {
    lfa::OtsuScratch ws;
    return lfa::otsu_float256(cnt, vmin, vmax, centre, &ws);
}

double hc_linspace_at(double a, double b, int n, int i) { return lfa::Linspace(a, b, n)(i); }

}
