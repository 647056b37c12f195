// A compiled host for the filter step, on the C ABI alone (include/eps_abi.h + the HIP runtime; no torch, no Python):
// what filter.py:96-142 + :160-161 do for `--model adamic_ogb --keep_top K` on a unit-valued symmetric graph.
//
//   filter_step graph.bin K out.bin
//
// graph.bin: int64 n, int64 nnz, int64 rowptr[n+1], int32 col[nnz]   (CSR, rows ascending, symmetric, no values)
// out.bin  : int64 rows, then rows x (int64 key = v << 32 | u, float score): the K best proposals (u, v), score descending,
//            then candidate order (column-major) ascending -- the order tests/test_gpu_examples.py compares with scan.scan_topk.
//
// Small graphs only: every column is scanned with no bar (the Python host estimates a bar from a column sample first:
// edge-proposal-sets_amd/scan.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "../include/eps_abi.h"

#define HIP_OK(x)                                                                       \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define EPS_OK_(x)                                                                      \
    do {                                                                                \
        if ((x) != 0) {                                                                 \
            fprintf(stderr, "%s: %s\n", #x, eps_last_error());                          \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

template <typename T>
static T *dev_alloc(size_t n)
{
    void *p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    return (T *)p;
}

int main(int argc, char **argv)
{
    if (argc != 4) {
        fprintf(stderr, "usage: %s graph.bin K out.bin\n", argv[0]);
        return 1;
    }
    const int64_t K = atoll(argv[2]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int64_t n = 0, nnz = 0;
    if (fread(&n, 8, 1, f) != 1 || fread(&nnz, 8, 1, f) != 1) return 1;
    std::vector<int64_t> rowptr(n + 1);
    std::vector<int32_t> col(nnz);
    if (fread(rowptr.data(), 8, n + 1, f) != (size_t)(n + 1) || fread(col.data(), 4, nnz, f) != (size_t)nnz) return 1;
    fclose(f);

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    int64_t *d_rowptr = dev_alloc<int64_t>(n + 1);
    int32_t *d_col = dev_alloc<int32_t>(nnz);
    HIP_OK(hipMemcpyAsync(d_rowptr, rowptr.data(), (n + 1) * 8, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice, stream));

    // weight prologue of adamic_utils.py:15-17: w[x] = 1 / log(colsum[x]), inf -> 0
    double *d_sum64 = dev_alloc<double>(n);
    float *d_sum = dev_alloc<float>(n), *d_w = dev_alloc<float>(n);
    EPS_OK_(eps_col_sums(d_rowptr, d_col, nullptr, n, n, d_sum64, d_sum, stream));
    EPS_OK_(eps_node_weights(d_sum, n, EPS_W_AA, d_w, stream));

    // per-graph tables of the threshold scan
    int32_t *d_revpos = dev_alloc<int32_t>(nnz);
    int64_t *d_fixw = dev_alloc<int64_t>(n);
    EPS_OK_(eps_reverse_positions(d_rowptr, d_col, n, d_revpos, nullptr, nullptr, stream));
    EPS_OK_(eps_fixed_weights(d_w, n, d_fixw, stream));
    int64_t win_ids = 0, n_win = 0;
    EPS_OK_(eps_filter_scan_windows(n, &win_ids, &n_win));
    int32_t *d_splits = nullptr;
    if (n_win > 1) {
        d_splits = dev_alloc<int32_t>((size_t)(n_win - 1) * n);
        EPS_OK_(eps_row_window_splits(d_rowptr, d_col, n, win_ids, n_win, d_splits, stream));
    }
    int64_t max_degree = 0, half_paths = 0;
    std::vector<int64_t> work(n, 0);                       // two-hop half paths per column: the hand-out order, the list size
    for (int64_t v = 0; v < n; ++v) {
        max_degree = std::max(max_degree, rowptr[v + 1] - rowptr[v]);
        for (int64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
            const int32_t w = col[e];
            work[v] += std::lower_bound(col.begin() + rowptr[w], col.begin() + rowptr[w + 1], (int32_t)v) - (col.begin() + rowptr[w]);
        }
        half_paths += work[v];
    }
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return work[a] > work[b]; });
    int32_t *d_order = dev_alloc<int32_t>(n);
    HIP_OK(hipMemcpyAsync(d_order, order.data(), n * 4, hipMemcpyHostToDevice, stream));

    // one launch over all columns, no bar: every candidate is a survivor
    const int64_t capacity = 2 * half_paths + 8192 * 320;
    if (capacity >= (1ll << 32)) {
        fprintf(stderr, "graph too large for the no-bar example\n");
        return 1;
    }
    int64_t *d_key = dev_alloc<int64_t>(capacity);
    float *d_val = dev_alloc<float>(capacity);
    HIP_OK(hipMemsetAsync(d_key, 0xff, capacity * 8, stream));                   // untouched slots stay -1
    eps_survivors rec;
    memset(&rec, 0, sizeof rec);
    rec.threshold = -__builtin_inff();
    rec.capacity = (uint32_t)capacity;
    rec.key = d_key;
    rec.val = d_val;
    eps_survivors *d_rec = dev_alloc<eps_survivors>(1);
    HIP_OK(hipMemcpyAsync(d_rec, &rec, sizeof rec, hipMemcpyHostToDevice, stream));
    const int64_t ws_bytes = eps_filter_scan_workspace_bytes(max_degree);
    void *d_ws = dev_alloc<char>(ws_bytes);
    EPS_OK_(eps_filter_scan(d_rowptr, d_col, d_revpos, d_fixw, d_splits, n, nnz, max_degree, d_order, n, d_rec, d_ws, ws_bytes,
                            stream));
    HIP_OK(hipMemcpyAsync(&rec, d_rec, sizeof rec, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (rec.count > rec.capacity) {
        fprintf(stderr, "survivor list overflow\n");
        return 4;
    }
    // the survivors, compacted on the device (slots are handed out in chunks: untouched ones kept their -1)
    const int64_t slots = rec.count;
    int64_t *d_ck = dev_alloc<int64_t>(slots), *d_m = dev_alloc<int64_t>(1);
    float *d_cv = dev_alloc<float>(slots);
    void *d_cws = dev_alloc<char>(eps_select_topk_cut_workspace_bytes());
    EPS_OK_(eps_compact_survivors(d_key, d_val, slots, d_ck, d_cv, d_m, d_cws, eps_select_topk_cut_workspace_bytes(), stream));
    int64_t m = 0;
    HIP_OK(hipMemcpyAsync(&m, d_m, 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if ((unsigned long long)m != rec.n_candidates) {
        fprintf(stderr, "survivors %lld != candidates %llu\n", (long long)m, rec.n_candidates);
        return 5;
    }

    // the K best directed rows in the declared order
    int64_t rows = 0;
    std::vector<int64_t> out_key;
    std::vector<float> out_val;
    if (m && K > 0) {
        int64_t *d_sk = dev_alloc<int64_t>(m), *d_nsel = dev_alloc<int64_t>(1);
        float *d_sv = dev_alloc<float>(m);
        EPS_OK_(eps_select_topk_cut(d_ck, d_cv, (int64_t)m, K, d_sk, d_sv, d_nsel, d_cws, eps_select_topk_cut_workspace_bytes(), stream));
        int64_t n_sel = 0;
        HIP_OK(hipMemcpyAsync(&n_sel, d_nsel, 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        rows = std::min<int64_t>(K, 2 * n_sel);
        int id_bits = 1;
        while ((1ll << id_bits) < n) ++id_bits;
        int64_t *d_ok = dev_alloc<int64_t>(rows);
        float *d_ov = dev_alloc<float>(rows);
        const int64_t rws = eps_select_topk_rows_workspace_bytes(n_sel);
        void *d_rws = dev_alloc<char>(rws);
        EPS_OK_(eps_select_topk_rows(d_sk, d_sv, n_sel, K, id_bits, d_ok, d_ov, d_rws, rws, stream));
        out_key.resize(rows);
        out_val.resize(rows);
        HIP_OK(hipMemcpyAsync(out_key.data(), d_ok, rows * 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(out_val.data(), d_ov, rows * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
    FILE *o = fopen(argv[3], "wb");
    if (!o) return 1;
    fwrite(&rows, 8, 1, o);
    for (int64_t i = 0; i < rows; ++i) {
        fwrite(&out_key[i], 8, 1, o);
        fwrite(&out_val[i], 4, 1, o);
    }
    fclose(o);
    printf("filter_step: %lld nodes, %lld candidates (unordered), %lld rows written\n", (long long)n, (long long)m, (long long)rows);
    return 0;
}
