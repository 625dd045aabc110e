// host_check — exercises gridding.hpp end to end on one GPU and prints checksums that
// tests/test_gpu_cpp_host.py compares with the CPU oracle on the same generated inputs.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "gridding.hpp"

using namespace gridding;

// tiny LCG so the Python test can regenerate the identical inputs
struct Lcg {
    uint64_t s;
    double next() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; }
};

int main(int argc, char **argv)
{
    const Int N = 96, W = 3, Q = 4, S = 7, n = argc > 1 ? atoll(argv[1]) : 4000;
    Lcg r{12345};
    WKernels gcf;
    gcf.W = W; gcf.Q = Q; gcf.gh = gcf.gw = S;
    gcf.data.resize((size_t)W * Q * Q * S * S);
    for (auto &z : gcf.data) { double a = r.next() - 0.5, b = r.next() - 0.5; z = {a, b}; }
    BaseLines p;
    std::vector<Int> wbin(n);
    std::vector<Visibility> vis(n);
    p.u.resize(n); p.v.resize(n); p.w.resize(n);
    for (Int k = 0; k < n; ++k) {
        p.u[k] = (r.next() - 0.5) * 1.1;
        p.v[k] = (r.next() - 0.5) * 1.1;
        p.w[k] = 0.0;
        wbin[k] = (Int)(r.next() * W) % W;
        double a = r.next() - 0.5, b = r.next() - 0.5;
        vis[k] = {a, b};
    }
    try {
        Backend be(0);
        Matrix<Visibility> a(N, N);
        Matrix<Visibility> g = be.convgrid2(gcf, a, p, wbin, vis);
        std::vector<Visibility> d = be.degrid2(gcf, g, p, wbin);
        Visibility gs = 0, ds = 0;
        double gabs = 0;
        for (Int y = 0; y < N; ++y)
            for (Int x = 0; x < N; ++x) { gs += g(y, x) * (double)(1 + (y * 31 + x * 17) % 7); gabs += std::abs(g(y, x)); }
        for (Int k = 0; k < n; ++k) ds += d[k] * (double)(1 + k % 5);
        printf("convgrid2 %.17g %.17g %.17g\n", gs.real(), gs.imag(), gabs);
        {   // the same through the node interface (a communicator over the devices given; here one)
            Node node(1);
            Matrix<Visibility> gn = node.convgrid2(gcf, a, p, wbin, vis);
            double diff = 0, mx = 0;
            for (size_t i = 0; i < gn.data.size(); ++i) {
                diff = std::max(diff, std::abs(gn.data[i] - g.data[i]));
                mx = std::max(mx, std::abs(g.data[i]));
            }
            printf("node %d %.3g\n", node.devices(), diff / mx);
        }
        printf("degrid2 %.17g %.17g\n", ds.real(), ds.imag());
        // error behaviour: a bad shape is reported, not silently ignored
        try {
            WKernels bad = gcf;
            bad.Q = 0;
            be.convgrid2(bad, a, p, wbin, vis);
            printf("error none\n");
        } catch (const Error &e) {
            printf("error %d\n", e.code);
        }
    } catch (const Error &e) {
        fprintf(stderr, "gridding error %d: %s\n", e.code, e.what());
        return 2;
    }
    return 0;
}
