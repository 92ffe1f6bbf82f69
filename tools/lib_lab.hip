// lib_lab: times the SHIPPED kernels (the kernel files are included verbatim) under different launch
// geometries, interleaved in one process.  Development tool, not part of the product.
#include "../deeppreconditioning_amd/csrc/dpcg_spmv.hip"
#include "../deeppreconditioning_amd/csrc/dpcg_pcg.hip"
#include "../deeppreconditioning_amd/csrc/dpcg_setup.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace dpcg;
namespace dpcg { void set_error(const std::string &) {} int hip_fail(hipError_t, const char *, const char *, int) { return -2; } }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 11;
    struct Case { int dim; int64_t n; };
    std::vector<Case> cases = {{3, 100}, {3, 256}};
    if (argc > 3) cases = {{atoi(argv[2]), atol(argv[3])}};
    std::vector<int> tile_grids = {1024, 1280, 1536};
    if (argc > 4) { tile_grids.clear(); for (int i = 4; i < argc; ++i) tile_grids.push_back(atoi(argv[i])); }
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto cs : cases) {
        const int64_t n2 = cs.n * cs.n, N = cs.dim == 2 ? n2 : n2 * cs.n;
        const int64_t nnz = cs.dim == 2 ? 5 * n2 - 4 * cs.n : 7 * n2 * cs.n - 6 * n2;
        CsrDev A; A.n = N; A.nnz = nnz;
        double *x, *y, *z, *r, *q, *dinv, *part_pq, *part_rz, *part_rr, *hist; Scalars *sc;
        CK(hipMalloc(&A.rowptr, (N + 1) * 4)); CK(hipMalloc(&A.col, nnz * 4)); CK(hipMalloc(&A.val, nnz * 8));
        for (double **p : {&x, &y, &z, &r, &q, &dinv}) CK(hipMalloc(p, N * 8));
        CK(hipMalloc(&part_pq, 4096 * 8)); CK(hipMalloc(&part_rz, 4096 * 8)); CK(hipMalloc(&part_rr, 4096 * 8));
        CK(hipMalloc(&hist, 4096 * 8)); CK(hipMalloc(&sc, sizeof(Scalars)));
        launch_gen_poisson(cs.dim, cs.n, A.rowptr, A.col, A.val, DPCG_F64, s);
        std::vector<double> hx(N); srand(1); for (auto &v : hx) v = (double)rand() / RAND_MAX - 0.5;
        for (double *p : {x, y, z, r, q, dinv}) CK(hipMemcpy(p, hx.data(), N * 8, hipMemcpyHostToDevice));
        CK(hipMemset(part_rr, 0, 4096 * 8)); CK(hipMemset(part_rz, 0, 4096 * 8)); CK(hipMemset(part_pq, 0, 4096 * 8));
        const double one = 1.0; CK(hipMemcpy(part_rr, &one, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(part_pq, &one, 8, hipMemcpyHostToDevice));
        launch_finalize_init(sc, part_rr, part_rr, part_rr, 1, 0.0, 0.0, hist, 0, nullptr, s);
        CK(hipStreamSynchronize(s));
        const double b_spmv = (double)nnz * 12 + (N + 1) * 4.0 + 16.0 * N;
        printf("== poisson%dd n=%ld N=%ld: SpMV %.1f MB, K2 %.1f MB, K3 %.1f MB\n", cs.dim, (long)cs.n, (long)N, b_spmv / 1e6, 40.0 * N / 1e6, 40.0 * N / 1e6);
        struct V { const char *name; int kind; int grid; bool ctl; double bytes; int npart; };
        std::vector<V> vs;
        for (int g : {1536, 2048}) { vs.push_back({"spmv+dot ctl (gather)", 0, g, true, b_spmv, 512}); }
        for (int g : tile_grids) { vs.push_back({"spmv+dot ctl (x-tile)", 3, g, true, b_spmv, 512}); }
        // x-tile plan
        SpmvPlan tplan; tplan.kernel = SPMV_TILE; tplan.nrb = (int)((N + 255) / 256);
        CK(hipMalloc(&tplan.tile_chunks, (size_t)tplan.nrb * kTileMaxChunks * 4)); CK(hipMalloc(&tplan.tile_nchunks, (size_t)tplan.nrb * 4)); CK(hipMalloc(&tplan.tile_lidx, nnz * 2 + 4));
        { int *fl; CK(hipMalloc(&fl, 8)); int hf[2] = {1, 0}; CK(hipMemcpy(fl, hf, 8, hipMemcpyHostToDevice));
          launch_tile_plan(A, tplan.nrb, tplan.tile_chunks, tplan.tile_nchunks, tplan.tile_lidx, fl, s); CK(hipStreamSynchronize(s));
          CK(hipMemcpy(hf, fl, 8, hipMemcpyDeviceToHost)); tplan.tile_max_chunks = hf[1]; printf("  tile plan: ok=%d max chunks=%d (LDS %zu B per block)\n", hf[0], hf[1], (size_t)(hf[1] * 64 + 2052) * 8); hipFree(fl); }
        vs.push_back({"K2 update_r np=2048", 1, 512, false, 40.0 * N, 2048});
        vs.push_back({"K3 update_xp np=512", 2, 512, false, 40.0 * N, 512});
        const int reps = N > 4000000 ? 10 : 50;
        std::vector<std::vector<float>> times(vs.size());
        for (int rd = 0; rd < rounds; ++rd)
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                const V &v = vs[vi];
                SpmvPlan plan; plan.kernel = SPMV_STREAM; plan.nrb = (int)((N + 255) / 256); plan.grid = std::min(plan.nrb, v.grid);
                IterCtl ctl{sc};
                auto go = [&]() {
                    if (v.kind == 3) { SpmvPlan tp = tplan; tp.grid = std::min(tp.nrb, v.grid); launch_spmv(A, tp, x, y, part_pq, v.ctl ? &ctl : nullptr, s); }
                    else if (v.kind == 0) launch_spmv(A, plan, x, y, part_pq, v.ctl ? &ctl : nullptr, s);
                    else if (v.kind == 1) launch_update_r(1, N, sc, part_pq, v.npart, q, r, dinv, z, part_rz, part_rr, v.grid, s);
                    else launch_update_xp(N, sc, part_rz, part_rr, v.npart, z, y, x, nullptr, hist, 0, v.grid, s);
                };
                // keep the state benign: K2 overwrites part_rr/part_rz, restore the never-converging control block
                go();
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) go();
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                times[vi].push_back(ms * 1e3f / reps);
                CK(hipMemsetAsync(part_rr, 0, 4096 * 8, s)); CK(hipMemcpyAsync(part_rr, &one, 8, hipMemcpyHostToDevice, s));
                CK(hipMemsetAsync(part_pq, 0, 4096 * 8, s)); CK(hipMemcpyAsync(part_pq, &one, 8, hipMemcpyHostToDevice, s));
                launch_finalize_init(sc, part_rr, part_rr, part_rr, 1, 0.0, 0.0, hist, 0, nullptr, s);
                for (double *p : {x, y, z, r, q}) CK(hipMemcpyAsync(p, dinv, N * 8, hipMemcpyDeviceToDevice, s));
            }
        {   // the iteration as the PCG loop runs it: K1 -> K2 -> K3, back to back
            for (int gs : {1536, 2048, -1536}) {
                SpmvPlan plan; plan.kernel = SPMV_STREAM; plan.nrb = (int)((N + 255) / 256); plan.grid = std::min(plan.nrb, gs);
                if (gs < 0) { plan = tplan; plan.grid = std::min(plan.nrb, -gs); }   // negative: the x-tile kernel
                IterCtl ctl{sc};
                std::vector<float> tt;
                for (int rd = 0; rd < rounds; ++rd) {
                    CK(hipMemsetAsync(part_rr, 0, 4096 * 8, s)); CK(hipMemcpyAsync(part_rr, &one, 8, hipMemcpyHostToDevice, s));
                    launch_finalize_init(sc, part_rr, part_rr, part_rr, 1, 0.0, 0.0, hist, 0, nullptr, s);
                    for (double *p : {x, y, z, r, q}) CK(hipMemcpyAsync(p, dinv, N * 8, hipMemcpyDeviceToDevice, s));
                    CK(hipEventRecord(e0, s));
                    const int vg = getenv("LAB_VEC_GRID") ? atoi(getenv("LAB_VEC_GRID")) : 512;
                    for (int i = 0; i < 40; ++i) {
                        launch_spmv(A, plan, y, q, part_pq, &ctl, s);
                        if (!getenv("LAB_STORE_Z")) {   // as the library runs Jacobi: z never stored (LAB_STORE_Z=1: the old form)
                            launch_update_r(1, N, sc, part_pq, plan.grid, q, r, dinv, z, part_rz, part_rr, vg, s, 0);
                            launch_update_xp(N, sc, part_rz, part_rr, vg, r, y, x, nullptr, hist, 0, vg, s, dinv);
                        } else {
                            launch_update_r(1, N, sc, part_pq, plan.grid, q, r, dinv, z, part_rz, part_rr, vg, s);
                            launch_update_xp(N, sc, part_rz, part_rr, vg, z, y, x, nullptr, hist, 0, vg, s);
                        }
                    }
                    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt.push_back(ms * 1e3f / 40);
                }
                std::sort(tt.begin(), tt.end());
                printf("  PIPELINE K1->K2->K3 spmv grid %d: median %8.2f us / iteration, min %8.2f\n", gs, tt[tt.size() / 2], tt[0]);
            }
        }
        for (size_t vi = 0; vi < vs.size(); ++vi) {
            auto &tv = times[vi]; std::sort(tv.begin(), tv.end());
            printf("  %-24s grid %4d  median %8.2f us  min %8.2f us  %7.1f GB/s\n", vs[vi].name, vs[vi].grid, tv[tv.size() / 2], tv[0], vs[vi].bytes / tv[tv.size() / 2] / 1e3);
        }
        hipFree(A.rowptr); hipFree(A.col); hipFree(A.val);
        for (double *p : {x, y, z, r, q, dinv, part_pq, part_rz, part_rr, hist}) hipFree(p);
        hipFree(sc);
    }
    return 0;
}
