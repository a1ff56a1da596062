// Probe (MI355X): where one node's ~13 000 cycles go inside the one-launch cell backward.  Includes a copy of csrc/stc_cell_bwd_x3.hip with
// s_memtime stamps at its phase boundaries (written by the recipe in DESIGN.md; stamps cost ~10 % themselves), runs it on random planes of
// the bench's size and prints cycles per node and phase, averaged over all waves.
//   python3 - <<< "see tools/probes/README"   (the stamped source is generated, not committed)
#include STAMPED_SOURCE
#include <cstdio>
#include <vector>
namespace stc { char* error_buffer() { static thread_local char b[512]; return b; } }
int main() {
    const long long nodes = 5ll * 50176; const int C = 32, h = 16, Lw = 32;
    const size_t plane = (size_t)nodes * C * h;
    std::vector<float> host(plane);
    for (size_t i = 0; i < plane; ++i) host[i] = 0.25f + 0.5f * ((i * 2654435761u) % 1000) / 1000.f;
    float* p[16];
    for (int i = 0; i < 13; ++i) { hipMalloc(&p[i], plane * 4); hipMemcpy(p[i], host.data(), plane * 4, hipMemcpyHostToDevice); }
    float *Tc, *Wg, *Wc, *pg, *pc;
    hipMalloc(&Tc, 2 * C * C * 4); hipMalloc(&Wg, 4 * Lw * 32 * 4); hipMalloc(&Wc, 4 * Lw * 16 * 4);
    hipMemcpy(Tc, host.data(), 2 * C * C * 4, hipMemcpyHostToDevice); hipMemcpy(Wg, host.data(), 4 * Lw * 32 * 4, hipMemcpyHostToDevice); hipMemcpy(Wc, host.data(), 4 * Lw * 16 * 4, hipMemcpyHostToDevice);
    hipMalloc(&pg, 512 * (4 * 32 * 32 + 32) * 4); hipMalloc(&pc, 512 * (4 * 32 * 16 + 16) * 4);
    int n_parts = 0;
    for (int rep = 0; rep < 3; ++rep) {
        unsigned long long zero[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zero, sizeof(zero));
        const int rc = stc_cell_bwd_planar_x3(p[0], p[1], p[2], p[3], Tc, Wg, Wc, p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], p[12], pg, pc, &n_parts, 1, 1, 0, 0, nodes, C, Lw, 0);
        hipDeviceSynchronize();
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamps), sizeof(st));
        const double n = (double)st[6];
        const char* names[6] = {"issue the next node's loads", "candidate convolution", "gate + blend backward", "gates: dG split, T_1 products, dZ tiles", "gates: T_1 dG, dWg", "loop end: wait for the prefetch, copy"};
        double tot = 0; for (int i = 0; i < 6; ++i) tot += st[i] / n;
        printf("rc %d, %llu waves, %.0f nodes, %.0f cycles per node\n", rc, st[7], n, tot);
        for (int i = 0; i < 6; ++i) printf("  %-42s %8.0f cycles  %5.1f %%\n", names[i], st[i] / n, 100.0 * st[i] / n / tot);
    }
    return 0;
}
