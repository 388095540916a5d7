// Unit-test shim: compiles the sequential device functions of image-analysis_amd/csrc/lfa_algo.h for the HOST so their
// arithmetic can be checked against the golden vectors on a machine without a GPU.  Test infrastructure only: nothing
// in the product loads this library.
#include <vector>
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

// threshold_otsu's bin search on an integer histogram with float32-rounded weights: returns the bin b (run end) of the
// first maximum, or -1.  Whole array as one tile.
int hc_otsu_int_seq(const uint32_t* cnt, const int32_t* lv, int nb) {
    std::vector<int32_t> f(lv, lv + nb), bk(nb + 1);
    f.push_back(-1);
    bk[0] = -1;
    for (int b = 0; b < nb; ++b) bk[b + 1] = lv[b];
    std::vector<float> w1(nb), w2(nb);
    std::vector<double> s1(nb), s2(nb);
    lfa::OtsuSeq a{0.0f, 0.0, 0}, z{0.0f, 0.0, 0};
    lfa::otsu_seq_fwd(cnt, f.data(), nb, a, w1.data(), s1.data());
    lfa::otsu_seq_bwd(cnt, bk.data(), nb, z, w2.data(), s2.data());
    int best = -1;
    double bv = -1.0;
    for (int b = 0; b + 1 < nb; ++b)
        if (lv[b + 1] != lv[b]) {
            const double v = lfa::otsu_seq_variance(w1[b], s1[b], w2[b + 1], s2[b + 1]);
            if (best < 0 || v > bv) { bv = v; best = b; }
        }
    return best;
}

double hc_linspace_at(double a, double b, int n, int i) { return lfa::Linspace(a, b, n)(i); }

}
