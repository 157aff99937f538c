/*
 * spurfies_hip.h — C ABI of libspurfies_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-ray volumetric-rendering hot path of kevinYitshak/spurfies.
 * Every entry point replaces a reference interface, cited as <file>:<line> relative to the
 * reference tree.  The reference binds its native op from Python (torch_knnquery, a CUDA
 * extension); this library is bound the same way, through ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types in signatures.
 *   - Every data pointer is a DEVICE pointer owned by the caller (a torch tensor's data_ptr());
 *     the library never frees or retains them beyond the call, except spf_grid_build, which
 *     copies what it needs into tables owned by the opaque spf_grid handle.
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it; no call synchronises the device unless it says so.
 *   - Return value: 0 on success, negative SPF_E* code otherwise; text in spf_last_error()
 *     (thread-local).  No exceptions cross the ABI.
 *   - All floating-point data is float32 (the reference forces fp32, spurfies/train.py:23-24),
 *     indices are int32.
 */
#ifndef SPURFIES_HIP_H
#define SPURFIES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPF_OK 0
#define SPF_EINVAL (-22)
#define SPF_ENOMEM (-12)
#define SPF_EHIP (-5)

#define SPF_ABI_VERSION 6
#define SPF_KMAX 8          /* neighbours per point (config/vol/dtu_pn.yaml:27, k: 8) */
#define SPF_GEO_DIM 32      /* geometry latent width = feature_vector_size/2 (pointneus_disent.py:172) */
#define SPF_COL_DIM 64      /* colour latent width  = feature_vector_size   (pointneus_disent.py:161) */
#define SPF_HID 256         /* MLP width (pointneus_disent.py:76-107) */
#define SPF_TILE_PTS 8      /* points per MLP tile (8 points x 8 neighbours = 64 MFMA rows) */

int spf_abi_version(void);
const char* spf_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Voxel grid  — replaces torch_knnquery.VoxelGrid
 *   ctor          spurfies/model/pointneus_disent.py:45-62
 *   set_pointset  spurfies/model/pointneus_disent.py:252-260,353-361,427-435,522-530,627-635
 *   query         spurfies/model/utils.py:93-95,118-120
 * Specification (frozen by this build; upstream source absent): see DESIGN.md §kNN.
 * ---------------------------------------------------------------------------------------- */
typedef struct spf_grid spf_grid;

typedef struct spf_grid_config {
    float voxel_size[3];           /* (0.025, 0.025, 0.025) */
    int32_t voxel_scale[3];        /* (3, 3, 3)  -> cell = voxel_size * voxel_scale */
    int32_t kernel_size[3];        /* (3, 3, 3)  -> occupancy dilation box */
    int32_t max_points_per_voxel;  /* 26 in the reference; applied only with SPF_KNN_TRUNCATE */
    int32_t max_occ_voxels;        /* 20000 in the reference; applied only with SPF_KNN_TRUNCATE */
    float ranges[6];               /* (xmin, ymin, zmin, xmax, ymax, zmax) */
    int32_t compat;                /* 0 = the frozen specification (exact radius-limited kNN, nothing dropped); or-ed SPF_KNN_* switches */
} spf_grid_config;

/* Upstream-compatibility switches (what the absent torch_knnquery source is BELIEVED to do — unverified):
 *   SPF_KNN_TRUNCATE  capacity limits, made deterministic: a cell keeps its max_points_per_voxel LOWEST-INDEX points, the grid keeps
 *                     the max_occ_voxels cells whose lowest point index is smallest (the order one-thread-per-point kernels claim them
 *                     when points arrive in index order; upstream's atomics / wall-clock-seeded reservoir make its own result random);
 *                     dropped points are never returned and do not occupy cells.
 *   SPF_KNN_LAYERED   layer-by-layer search with early exit: if the sample's own cell already holds k points within the radius, those
 *                     k (nearest of that cell) are returned and the 26 surrounding cells are not searched. */
#define SPF_KNN_TRUNCATE 1
#define SPF_KNN_LAYERED 2

typedef struct spf_grid_info {
    float origin[3];
    float cell[3];
    int32_t dims[3];
    int32_t n_points;     /* points handed to spf_grid_build */
    int32_t n_in_range;   /* points inside `ranges` (the others are never returned) */
    int32_t n_occupied;   /* occupied cells */
    int32_t max_cell_points; /* points in the fullest cell (> max_points_per_voxel: upstream would drop some at random) */
} spf_grid_info;

int spf_grid_create(const spf_grid_config* cfg, spf_grid** out);
void spf_grid_destroy(spf_grid* g);

/* VoxelGrid.set_pointset.  points: [n,3].  Builds the cell table (counting sort) and the
 * dilated-occupancy bitmask.  Synchronises `stream` once (the bounding box sizes the tables).
 * The reference rebuilds the grid 3-4x per step for a constant cloud; callers of this library
 * build once per cloud. */
int spf_grid_build(spf_grid* g, const float* points, int32_t n, void* stream);
int spf_grid_get_info(const spf_grid* g, spf_grid_info* out);

/* VoxelGrid.query, dense (uncompacted) form.  raypos: [R,D,3].
 *   pidx        [R,SR,k] int32   neighbours sorted by (dist2, index), -1 padded
 *   loc         [R,SR,3] float   position of the sample feeding each slot (0 where unused)
 *   slot_sample [R,SR]   int32   sample index (0..D-1) feeding the slot, -1 = unused slot
 *   slot_valid  [R,SR]   uint8   1 iff the slot has >= 1 neighbour ("valid point", utils.py:98-101)
 *   ray_valid   [R]      uint8   1 iff some slot of the ray is valid (the op's ray_mask)
 * radius_limit_scale multiplies max(voxel_size_x, voxel_size_y) as upstream does.  k <= SPF_KMAX. */
int spf_grid_query(const spf_grid* g, const float* raypos, int32_t R, int32_t D, int32_t k,
                   float radius_limit_scale, int32_t SR, int32_t* pidx, float* loc,
                   int32_t* slot_sample, uint8_t* slot_valid, uint8_t* ray_valid, void* stream);

/* ABI 6 — front end of the mesh-extraction sweep (spurfies/utils/plots.py:249-253: get_sdf_eval over the points of a regular grid, chunk by
 * chunk): the grid np.meshgrid(xs, ys, zs) ('xy' indexing, raveled: point i = (xs[(i / nz) % nx], ys[i / (nx nz)], zs[i % nz])) is never
 * materialised.  For the flat indices [first, first + count): a point that fails the dilated-occupancy test spf_grid_query starts with gets
 * fill[i - first] = fill_value (its SDF is the 1000 filler whatever else happens); the others leave compacted, in no particular order:
 * pts[n,3], idx[n] = i, with n ADDED to counter[0] (device, zero before the first call of a sweep; pts / idx must hold the running total).
 * The caller evaluates spf_grid_query / spf_geo_forward on pts and scatters the results to idx. */
int spf_grid_sweep_hits(const spf_grid* grid, const float* xs, const float* ys, const float* zs, int32_t nx, int32_t ny, int32_t nz,
                        int64_t first, int64_t count, float* fill, float fill_value, float* pts, int64_t* idx, uint64_t* counter, void* stream);

/* ABI 5: the neighbour search of spf_grid_query alone, for slots that were assigned elsewhere: slot_sample [R,SR] (INPUT: sample index of
 * each slot or -1, a ray's first SR dilated-occupancy hits in sample order) and ray_valid [R] cleared to 0 come from the caller —
 * spf_sampler_train assigns the slots while it still holds the ray's sorted samples.  Same outputs as spf_grid_query. */
int spf_grid_knn(const spf_grid* g, const float* raypos, int32_t R, int32_t D, int32_t k, float radius_limit_scale, int32_t SR,
                 const int32_t* slot_sample, int32_t* pidx, float* loc, uint8_t* slot_valid, uint8_t* ray_valid, void* stream);

/* Device-side compaction of valid points (replaces the masked_select host syncs of
 * spurfies/model/utils.py:107-112 and mask_to_batch_ray_idx :172-183).
 *   point_slot [R*SR] int32  flat slot id (r*SR+s) of the p-th valid point, ray-major order
 *   slot_point [R*SR] int32  inverse map, -1 for slots that are not valid points
 *   n_points   [1]    int32  number of valid points P (stays on the device)
 * scratch: >= R+1 int32.
 * fill_sdf [R*SR] / fill_grad [R*SR,3] (optional): every slot of them is set to fill_value / 0 on the way — the rows spf_geo_forward
 * leaves untouched because they are not valid points (the reference's 1000 filler, pointneus_disent.py:271,371,445,703). */
int spf_compact_points(const uint8_t* slot_valid, int32_t R, int32_t SR, int32_t* point_slot,
                       int32_t* slot_point, int32_t* n_points, int32_t* scratch, float* fill_sdf, float fill_value,
                       float* fill_grad, void* stream);

/* Load-time voxel thinning of the .ply cloud (spurfies/model/utils.py:21-27, construct_vox_points_closest): integer cell of
 * every point, cells[i] = floor((xyz[i] - space_min) / voxel) in float32 with IEEE division (what torch computes on the CPU;
 * a device-side elementwise division need not be correctly rounded).  space_min: HOST float[3]; xyz [n,3], cells [n,3] device. */
int spf_voxel_cells(const float* xyz, int64_t n, const float* space_min, float voxel, int32_t* cells, void* stream);

/* Pair list over the valid points (replaces mask_to_batch_ray_idx, spurfies/model/utils.py:172-183): the rows
 * of the MLP kernels are the valid (point, neighbour) pairs, grouped by point, WITHOUT padding.
 *   pair_off   [max_points+1] int32  exclusive scan of the per-point neighbour counts (pair_off[P] = n_pairs)
 *   pair_point [max_points*k] int32  compact point id of each pair
 *   n_pairs    [1] int32 (device);   scratch: >= max_points/2048 + 2 int32 */
int spf_build_pairs(const int32_t* nbr, const int32_t* point_slot, const int32_t* n_points, int32_t max_points,
                    int32_t k, int32_t* pair_off, int32_t* pair_point, int32_t* n_pairs, int32_t* scratch,
                    void* stream);

/* spf_compact_points + spf_build_pairs in ONE pair of launches (what the model's passes use: the four-launch form costs more in
 * dispatch than in work at 10^5 slots): nbr [R*SR, k] is indexed by SLOT (the kNN output as it is); outputs as above, with
 * counts[2] = {n_points, n_pairs} on the device and the same optional fillers.  scratch: >= 2 * (R*SR / 2048 + 1) int32.
 * gate (DEVICE int32, may be NULL): when *gate == 0 the counts are reported as {0, 0} (the fillers are still laid down), so the MLP kernels
 * behind this pass do no work — how a sampler iteration the loop did not reach is skipped without a host round trip (spf_sampler_iter).
 * sync (DEVICE uint64, may be NULL; ABI 4): spf_compact_sync_words(R*SR) words that are ALL ZERO on entry (e.g. hipMemset once) and are left
 * all zero — the pass then runs as ONE launch in which 512-slot chunks publish their totals to each other through these words
 * (relaxed agent-scope atomics, value and flag in one word; the last chunk to have read clears them; the last chunk itself publishes
 * nothing, so every word that was written has provably been read before the clear).  The buffer must not be used by two launches that can
 * run at the same time (the launch spins on words its own blocks publish): one buffer per OWNER of a pass — a stream is not an owner
 * once launches are captured into graphs that replay anywhere.  Passes of more than 2048 chunks (1 M slots) and sync == NULL take the
 * two-launch form. */
int64_t spf_compact_sync_words(int64_t n_slots);
int spf_compact_pairs(const uint8_t* slot_valid, const int32_t* nbr, int32_t R, int32_t SR, int32_t k, int32_t* point_slot,
                      int32_t* slot_point, int32_t* pair_off, int32_t* pair_point, int32_t* counts, int32_t* scratch,
                      float* fill_sdf, float fill_value, float* fill_grad, const int32_t* gate, uint64_t* sync, void* stream);
/* ABI 5: the same pass with spf_filter_points (pointneus_disent.py:207-239) riding along in the same launch: loc [R*SR,3] (spf_grid_query),
 * cam_loc / ray_dirs [R,3] -> z, deltas [R*SR], x [R*SR,3], bit for bit spf_filter_points' values (one launch less on the main pass). */
int spf_compact_pairs_filter(const uint8_t* slot_valid, const int32_t* nbr, int32_t R, int32_t SR, int32_t k, int32_t* point_slot,
                             int32_t* slot_point, int32_t* pair_off, int32_t* pair_point, int32_t* counts, int32_t* scratch,
                             float* fill_sdf, float fill_value, float* fill_grad, const int32_t* gate, uint64_t* sync,
                             const float* loc, const float* cam_loc, const float* ray_dirs, float* z, float* deltas, float* x, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused geometry path — replaces get_keypoint_data + compute_weights + get_sdf (+ the value of
 * get_gradients): spurfies/model/utils.py:140-170, spurfies/model/pointneus_disent.py:241-247,
 * 300-323; the same code inlined in sdf_importance :386-418, pseudo_sdf :460-493,
 * get_sdf_eval :284-296.
 * ---------------------------------------------------------------------------------------- */

/* Arithmetic of the MLP / weight-gradient kernels, chosen PER CALL (`arith` argument; the library keeps no process-wide mode):
 *   SPF_ARITH_SPLIT  every fp32 operand is split into three bf16 pieces (x = p1 + p2 + p3 exactly to 24 bits) and the six piece
 *                    products with i + j <= 4 run on the bf16 matrix pipe with fp32 accumulation: each piece product is exact,
 *                    the three dropped ones are below 2^-24 of the product each — fp32-CLASS accuracy (<= 2 ulp per product,
 *                    tests/test_gpu_wgrad.py; results differ from SPF_ARITH_F32 by that and by summation order), at 2.7x the
 *                    matrix rate.  The product path's arithmetic of rounds 1 - 5 (since round 6 the Python layer passes SPF_ARITH_H2 where a kernel
 *                    takes it).  The geometry kernel runs it on v_mfma_f32_16x16x32_bf16.
 *   SPF_ARITH_F32    v_mfma_f32_32x32x2_f32 (verification twin).
 *   SPF_ARITH_SPLIT_W  spf_geo_forward only: SPF_ARITH_SPLIT's arithmetic on v_mfma_f32_32x32x16_bf16 tiles (same products, other
 *                    summation order) — the second MFMA shape of the dominant kernel, kept so that a benchmark can time both shapes
 *                    in one process on the box it runs on (bench.py: roofline.ab).
 * A backward must be called with the arith of its forward (the two keep LeakyReLU sign bits in different layouts). */
#define SPF_ARITH_SPLIT 0
#define SPF_ARITH_F32 1
#define SPF_ARITH_SPLIT_W 2
/* ABI 6 — "H2" arithmetic on v_mfma_f32_32x32x16_f16: every fp32 operand as TWO fp16 pieces (22 mantissa bits), an fp32 product = THREE exact fp16
 * piece products (the fourth is <= 2^-22 of the product and dropped) accumulated in fp32: three matrix instructions per product instead of six.
 * Per product |error| <= 3 x 2^-22 (bf16 x 3: ~2e-7; fp32: 6e-8); per OUTPUT of a 256-term layer the measured error against float64 equals the
 * fp32-MFMA kernels' (profiles/r06_engine_accuracy.json, r06_color_accuracy.json).  Accepted by
 *   spf_geo_forward                 x = h1 + 2^-11 h2; main + cross accumulators combined once per layer.  The same kernel as SPF_ARITH_SPLIT_W
 *                                   otherwise.  Activations must lie within fp16's range (|x| < 65504; below 6e-5 an absolute accuracy of 1.5e-11).
 *   spf_color_forward / _backward   a member of the SPF_ARITH_SPLIT family: same operand layouts, sign words and bias-gradient convention, so forward
 *                                   and backward may differ in it (SPLIT forward + H2 backward and vice versa are valid pairs).  The backward
 *                                   carries every gradient ROW with its own power of two (rows span 1 .. 1e-44 and zeros: all finite).
 *   spf_wgrad / spf_wgrad_batched   (C > 32) pieces h1 = fp16(s x), h2 = fp16(s x - h1) into ONE accumulator; A scaled by 4 (|A| < 16376), G per
 *                                   wave and 32-column block by a power of two chosen on the fly: a term is exact relative to the LARGEST
 *                                   terms of its block (csrc/wgrad.hip, tests/test_gpu_wgrad.py).
 *   spf_rhead_forward / _backward   as the colour pair: a member of the SPF_ARITH_SPLIT family, forward and backward independently; the backward carries
 *                                   every gradient row (= point) with its own power of two.
 * The narrow (C <= 32) weight-gradient kernel is fp32 MFMA whatever `arith` says.  A value outside fp16's range becomes inf / NaN — loudly:
 * the optimiser step's finite check (spf_adam_step) then skips and counts the update. */
#define SPF_ARITH_H2 3
/* spf_geo_forward only, OR-ed into arith: the bf16-piece kernels of THIS launch stamp the held-clock counters (spf_geo_clock_read).  Without
 * the bit a launch touches no state outside its arguments. */
#define SPF_ARITH_CLOCK 0x100
/* ABI 6 — spf_geo_forward only, OR-ed into SPF_ARITH_SPLIT_W for an SDF-ONLY pass (grad == jac == NULL): REDUCED products — two bf16 pieces
 * per operand, three piece products instead of six (~16 mantissa bits per product instead of fp32-class), half the matrix instructions.
 * OPT-IN for the evaluation sampler's SDF passes (ray_sampler.py:403: values that only steer where the next samples go; the main pass
 * re-evaluates every rendered point with the full products); never for a pass whose values are rendered, differentiated or reported.
 * Tolerance study: tests/test_gpu_sampler.py::test_reduced_product_sampler_passes..., DESIGN.md section 5. */
#define SPF_ARITH_LITE 0x200

/* Number of floats of the packed F_geometry/T weight image. */

int64_t spf_geo_packed_floats(void);

/* Pack nn.Linear weights ([out,in] row-major, as state_dict holds them) of
 * F_geometry.{0,2,4,6,8} and T.0 into the MFMA fragment order the kernels stream.
 * The last Linear of F_geometry and T are folded (no activation sits between them,
 * pointneus_disent.py:95-98): v = T.W * F8.W, c = T.W * F8.b + T.b. */
int spf_geo_pack(const float* w0, const float* b0, const float* w2, const float* b2,
                 const float* w4, const float* b4, const float* w6, const float* b6,
                 const float* w8, const float* b8, const float* wT, const float* bT,
                 float* packed, void* stream);

/* Evaluates every valid pair q < *n_pairs (tiles of 64 pairs) and reduces per point p < *n_points
 * (both DEVICE int32; max_points / max_pairs size the launches).  Per-point arrays are addressed by ROW:
 * row = point_slot[p] (the flat slot id r*SR+s written by spf_compact_points), or row = p when point_slot
 * is NULL; per-pair arrays by the compact pair id q (spf_build_pairs).  Inputs: x[row,3] positions,
 * nbr[row,k] int32 neighbours (-1 pad), tables pts[N,3], feat_geo[N,32], the packed weight image,
 * rbf = conf.rbf (45).  Outputs (rows of invalid points are left untouched: pre-fill sdf with 1000):
 *   sdf   [rows]         RBF-weighted mean of the per-neighbour SDF
 *   grad  [rows,3]       d sdf / d x  (NULL: skip the Jacobian sweep — sampler / eval mode)
 *   wn    [max_pairs]    normalised RBF weight of each pair, w_j / sum_j w_j (may be NULL when grad is NULL)
 *   jac   [max_pairs,32] d sdf_j / d latent_j per pair (NULL iff grad is NULL)
 *   pair_tmp [max_pairs,5] scratch: {w_j, sdf_j, d sdf_j / d x (3)} per pair.
 * ABI 5: sdf == NULL (only with grad == NULL) skips the per-point reduction — pair_tmp is then the result, reduced by the consumer
 * (spf_sampler_train forms sum_j w_j sdf_j / sum_j w_j per sample on its way, in the same order). */
int spf_geo_forward(const float* x, const int32_t* nbr, const int32_t* point_slot, const int32_t* pair_off,
                    const int32_t* pair_point, const int32_t* n_points, const int32_t* n_pairs,
                    int32_t max_points, int32_t max_pairs, int32_t k, const float* pts, const float* feat_geo,
                    const float* packed, float rbf, float* sdf, float* grad, float* wn, float* jac,
                    float* pair_tmp, int32_t arith, void* stream);

/* Diagnostics (no reference counterpart; bench.py's `roofline.held_clock`): the shader clock the chip held while the bf16-piece
 * geometry kernels ran since the last reset, measured by the kernels themselves (s_memtime / s_memrealtime stamps of every workgroup) —
 * only by launches that asked for it (spf_geo_forward's arith | SPF_ARITH_CLOCK).  The counters are the library's ONE piece of device-global
 * state: one array per device and process, shared by every stream and caller (two measuring callers see each other's launches; a reset
 * wipes both) and read from the CURRENT device — a diagnostic for a single measuring caller (bench.py), off by default.
 * out12 (HOST, 12 x uint64) = [mfma shape: 0 = 16x16x32 (SPF_ARITH_SPLIT), 1 = 32x32x16 (SPF_ARITH_SPLIT_W)][0 = without, 1 = with the
 * Jacobian sweep][shader cycles, 100 MHz ticks, workgroups], summed over workgroups; clock [GHz] = 0.1 * cycles / ticks.  Synchronous
 * (a device-to-host copy of the current device's counters); reset != 0 zeroes them afterwards. */
int spf_geo_clock_read(uint64_t* out12, int32_t reset);

/* Backward of the weighted mean w.r.t. the geometry latents:
 *   g_feat_geo[nbr(q), :] += g_sdf[row(q)] * wn[q] * jac[q,:]   (float atomics)
 * (F_geometry / T are frozen in the reference's training, train.py:151-154.)
 * Optional, in the same launch: the gradient w.r.t. the query positions, g_x[row,:] = g_sdf[row] * grad_x[row,:] for row < n_rows
 * (grad_x = spf_geo_forward's `grad`; the RBF weights are detached, pointneus_disent.py:242) — what autograd's backward of the
 * pseudo-point SDF needs (pointneus_disent.py:765-780); g_x NULL: skipped. */
int spf_geo_backward_latents(const float* g_sdf, const float* wn, const float* jac, const int32_t* nbr,
                             const int32_t* point_slot, const int32_t* pair_off, const int32_t* pair_point,
                             const int32_t* n_pairs, int32_t max_pairs, int32_t k, float* g_feat_geo,
                             int64_t* g_feat_geo_fixed, const float* grad_x, float* g_x, int32_t n_rows, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused colour-feature path — replaces the F_color half of get_color,
 * spurfies/model/pointneus_disent.py:325-336 (posenc embedder.py:26-30, gather utils.py:140-170,
 * GEMMs, index_add_), and autograd's backward through it.
 * F_color.6 — the last layer, linear (pointneus_disent.py:76-85) — commutes with the RBF-weighted mean
 * (sum_j wn_j (W6 a_j + b6) = W6 sum_j wn_j a_j + b6, the weights sum to 1), so this stage ends at the mean of the
 * third activation, agg3, and the head stage below applies F_color.6 once per POINT instead of once per pair.
 * ---------------------------------------------------------------------------------------- */
/* arith (see SPF_ARITH_*): with SPF_ARITH_SPLIT spf_color_backward leaves g_b0 / g_b2 / g_b4 untouched — the bias gradients are
 * the column sums of G1..G3 and come from spf_wgrad's dbias output; with SPF_ARITH_F32 spf_color_backward accumulates them. */

int64_t spf_color_packed_floats(void);

/* Pack F_color.{0,2,4} ([out,in] row-major) into forward and transposed fragment order.
 * zero_buf (may be NULL; 16-byte aligned) / zero_floats: a buffer cleared in the same launch — the agg3 accumulator spf_color_forward adds
 * into (the weights change every optimisation step, so this launch precedes every forward anyway). */
int spf_color_pack(const float* w0, const float* b0, const float* w2, const float* b2, const float* w4,
                   const float* b4, float* packed, float* zero_buf, int64_t zero_floats, void* stream);

/* agg3[p, 256] += sum_j wn[q] * a3_q,  a3 = lrelu(F_color.4(lrelu(F_color.2(lrelu(F_color.0([posenc6(x[row] - pts[nbr]) |
 * feat_color[nbr]])))))) over the pairs q of the p-th valid point (pair lists from spf_build_pairs, wn from
 * spf_geo_forward); agg3 must be ZEROED by the caller (a point whose pairs straddle two 64-pair tiles receives two
 * commutative atomic adds).  Training mode (act0 != NULL) also writes, per pair row q (T = 64 * ceil(n_pairs/64) rows):
 *   act0 [T,104]   layer-0 input in the kernel's internal column order [latent 64 | posenc 39 | 0]
 *   act1, act2 [T,256]  inputs of F_color.2 and F_color.4 (operands of their weight-gradient GEMMs)
 *   masks [T/64, 3, 512] uint32  LeakyReLU sign bits of the three layers in accumulator order */
int spf_color_forward(const float* x, const int32_t* nbr, const float* wn, const int32_t* point_slot,
                      const int32_t* pair_off, const int32_t* pair_point, const int32_t* n_pairs, int32_t max_pairs,
                      int32_t k, const float* pts, const float* feat_color, const float* packed, float* agg3,
                      float* act0, float* act1, float* act2, uint32_t* masks, int64_t* agg3_fixed,
                      int32_t arith, void* stream);

/* Data-gradient chain for g_agg3[p,256] = dL/d agg3: writes the pre-activation gradients G1..G3 [T,256]
 * (weight gradients of F_color.0/2/4 are then dW_l = G_l^T act_{l-1}: spf_wgrad), ADDS their column sums to
 * g_b0, g_b2, g_b4 [256] (bias gradients of F_color.0/2/4) and ADDS the colour-latent gradient into
 * g_feat_color[N,64] (float atomics).  The accumulating outputs may point straight into the gradient buffers. */
int spf_color_backward(const float* g_agg3, const int32_t* nbr, const float* wn, const int32_t* point_slot,
                       const int32_t* pair_off, const int32_t* pair_point, const int32_t* n_pairs, int32_t max_pairs,
                       int32_t k, const float* packed, const uint32_t* masks, float* G1, float* G2, float* G3,
                       float* g_b0, float* g_b2, float* g_b4, float* g_feat_color, int64_t* g_feat_color_fixed,
                       int32_t arith, void* stream);

/* ------------------------------------------------------------------------------------------
 * Head stage, per valid POINT (tiles of 64): agg = F_color.6(agg3), then the radiance head R — replaces F_color's last
 * layer and the second half of get_color, spurfies/model/pointneus_disent.py:333-346 (view encoding with multires 3,
 * concat, R: 277 -> 256 -> 256 -> 3, sigmoid) and its backward.
 * ---------------------------------------------------------------------------------------- */
/* arith (see SPF_ARITH_*): with SPF_ARITH_SPLIT spf_rhead_backward leaves g_b6 / g_b0 / g_b2 untouched — they are the column
 * sums of g_agg / G1 / G2 and come from spf_wgrad's dbias output; with SPF_ARITH_F32 spf_rhead_backward accumulates them. */

int64_t spf_rhead_packed_floats(void);
/* zero_buf / zero_floats: as spf_color_pack — here the dense colors [rows,3] array spf_rhead_forward scatters into (0 at invalid slots). */
int spf_rhead_pack(const float* w6, const float* b6, const float* w0, const float* b0, const float* w2, const float* b2,
                   const float* w4, const float* b4, float* packed, float* zero_buf, int64_t zero_floats, void* stream);

/* colors[row,3] = sigmoid(R([direnc3(ray_dirs[row / SR]) | W6 agg3[p] + b6])) for the p-th valid point, row = point_slot[p]
 * (rows of invalid points untouched: pre-fill with 0).  Training mode (direnc != NULL) stores, per point
 * (T = 64*ceil(P/64) rows): agg [T,256] (= F_color.6's output, R.0's input), direnc [T,24] (21 + 3 zeros),
 * act1, act2 [T,256], masks [T/64,2,512] (LeakyReLU sign words, 16 words per point; opaque: written and read back by this pair of entry
 * points only).  SPF_ARITH_SPLIT, round 5: the point list is worked off in whole rounds of 64-point tiles (one tile per workgroup and round) and
 * the remainder — less than one round — in 32-point tiles when that is at most one per workgroup; spf_rhead_backward splits the list at the
 * same place (same count, same max_points => same grid), which is what lets it read the sign words the forward stored. */
int spf_rhead_forward(const float* agg3, const float* ray_dirs, const int32_t* point_slot, const int32_t* n_points,
                      int32_t max_points, int32_t SR, const float* packed, float* colors, float* agg, float* direnc,
                      float* act1, float* act2, uint32_t* masks, int32_t arith, void* stream);

/* Given g_colors[row,3]: writes G1, G2 [T,256] (pre-activation gradients of R.0 / R.2; dW_l = G_l^T act_{l-1}: spf_wgrad),
 * g_agg [T,256] = dL/d agg (F_color.6's weight gradient is g_agg^T agg3) and g_agg3 [T,256] = g_agg W6 (input of
 * spf_color_backward); rows >= P are scratch.  ADDS into g_b6 [256] (F_color.6.bias), g_b0, g_b2 [256], g_w4 [3,256],
 * g_b4 [3] (R.0.bias, R.2.bias, R.4.weight, R.4.bias; float atomics — the caller zeroes them or points them at its
 * gradient buffers).  g_w4_fixed [768] / g_b4_fixed [3] (both or neither; SPF_ARITH_SPLIT): fixed-point accumulators (see "Reproducible
 * gradients" below) that receive the tiles' terms instead of g_w4 / g_b4. */
int spf_rhead_backward(const float* g_colors, const float* colors, const int32_t* point_slot, const int32_t* n_points,
                       int32_t max_points, const float* packed, const float* act2, const uint32_t* masks,
                       float* G1, float* G2, float* g_agg, float* g_agg3, float* g_b6, float* g_b0, float* g_b2,
                       float* g_w4, float* g_b4, int64_t* g_w4_fixed, int64_t* g_b4_fixed, int32_t arith, void* stream);

/* ------------------------------------------------------------------------------------------
 * VolSDF error-bounded sampler — replaces UniformSampler.get_z_vals (spurfies/model/ray_sampler.py:33-59),
 * ErrorBoundSampler_pn.get_z_vals (:377-574) and get_error_bound (:576-588).  The random numbers are
 * drawn by the host from the CPU generator, as the reference does, and passed in.
 * ---------------------------------------------------------------------------------------- */

/* z[r,i] = near (1 - tlin[i]) + far tlin[i]; with t_rand [R,n] (training) stratified inside the
 * mid-point intervals.  Also writes points[r,i,:] = cam_loc[r] + z ray_dirs[r]. */
int spf_sampler_uniform(const float* tlin, const float* t_rand, const float* cam_loc, const float* ray_dirs,
                        int32_t R, int32_t n, float near, float far, float* z, float* points, void* stream);

/* One refinement iteration on sorted z [R,n] with sdf [R,n] (1000 where a sample has no neighbour):
 * d* per interval, beta by bisection from beta_in [R] (NULL: the Lemma-2 bound sqrt(bound_coef * sum dz^2)),
 * beta0 a DEVICE scalar, then N samples by inverting the CDF of
 *   more == 0: the rendering weights + 1e-5           (final sample set, :491-503)
 *   more != 0: the per-interval error bound + add_tiny (:470-489), and z_merged [R,n+N] =
 *              sort(cat(z, samples)) with merged_idx [R,n+N] indexing into cat(z, samples).
 * u: [N] (u_per_ray == 0) or [R,N].  N may be 0 (beta only).  beta_out [R].
 * flags / it — DEVICE-SIDE LOOP CONTROL (flags NULL: off).  The reference decides after every iteration on the host whether another one follows
 * (`not_converge = beta.max() > beta0`, ray_sampler.py:468: one synchronisation per iteration).  Every iteration's shapes are static (n = 128 (it + 1)),
 * so a caller may instead enqueue ALL iterations with flags [>= max iterations + 1] int32, flags[0] = 1, the rest 0: flags[i] != 0 <=> the loop
 * reaches iteration i.  A call for iteration `it` returns at once unless flags[it] != 0; the beta-only call (N == 0) sets flags[it + 1] when some
 * ray's beta exceeds beta0; a merging call (more != 0) runs iff flags[it + 1] != 0, a final call (more == 0, N > 0) iff flags[it + 1] == 0 —
 * so exactly the calls the reference's loop would have made do work.  spf_sampler_finish takes the same pair and runs behind the final call;
 * spf_compact_pairs' gate = &flags[it] empties the SDF pass of an iteration that is not reached.  sum(flags) = iterations realised. */
int spf_sampler_iter(const float* z, const float* sdf, const float* beta_in, const float* beta0, int32_t R,
                     int32_t n, float eps, float bound_coef, int32_t beta_iters, int32_t more, float add_tiny,
                     const float* u, int32_t u_per_ray, int32_t N, float* samples, float* beta_out,
                     float* z_merged, int32_t* merged_idx, int32_t* flags, int32_t it, void* stream);

/* z_out [R,Ns+2+Ne] = sort(cat(z_samples [R,Ns], near, far, z_vals[:, sel])) and the main-pass points
 * points[r,i,:] = cam_loc[r] + z_out ray_dirs[r]  (:535-559). */
int spf_sampler_finish(const float* z_samples, int32_t Ns, const float* z_vals, int32_t n, const int32_t* sel,
                       int32_t Ne, float near, float far, const float* cam_loc, const float* ray_dirs, int32_t R,
                       float* z_out, float* points, const int32_t* flags, int32_t it, void* stream);

/* ABI 6 — ONE ITERATION of the evaluation sampler loop (ErrorBoundSampler_pn.get_z_vals with fast = -1, ray_sampler.py:397-533) behind the
 * geometry kernel's SDF-only launch over the iteration's NEW samples (spf_geo_forward with sdf == NULL), in two launches instead of ~14
 * (point reduce, cat, gather, beta pass, two sampling passes with three output fills, finish, index cast, two point ops, hit slots):
 *   phase 0  "test"  the iteration's SDF row sdf_cur [R,n]: entry k comes from the previous row (sdf_prev [R,n_prev]) or from the pair scratch
 *                    of the new samples, selected by merged_idx [R,n] (index into cat(previous z, new samples); n_prev == 0: all new);
 *                    beta [R] by bisection (ray_sampler.py:420-445); flags[it + 1] |= 1 if any ray's beta > beta0 (the convergence test :468).
 *   phase 1  "step"  flags[it + 1] decides.  Set: N_more samples from the error-bound pdf (u_more), z_merged / merged_out [R, n + N_more] (sorted
 *                    merge, :532-533) and points_new [R,N_more,3] = cam_loc + s ray_dirs, the next iteration's query points.  Clear: N_fin samples
 *                    from the weight pdf (u_fin), then spf_sampler_finish's z_out / points_out [R, N_fin + 2 + Ne(,3)] and the main pass's slot
 *                    assignment (slot_sample [R,SR], ray_valid [R] cleared: spf_grid_knn takes them) — and NaN in points_new.
 *   phase 2  "last"  the last iteration the loop may take (it == max_total_iters - 1): phase 0's input stage, the full bisection, then phase
 *                    1's final branch unconditionally (:434-445, no test).
 * Every launch returns at once when flags[it] == 0 (the loop ended earlier); all shapes are static, nothing synchronises.  u_more / u_fin
 * are shared by all rays (torch.linspace(0, 1, N), :509-511 in evaluation).  Same arithmetic, same order as the separate launches
 * (tests/test_gpu_sampler.py compares them bit for bit). */
typedef struct spf_sampler_eval_args {
    const float* z;              /* [R,n] sorted depths of this iteration */
    int32_t n, n_prev;
    const float* sdf_prev;       /* [R,n_prev] (n_prev > 0) */
    const int32_t* merged_idx;   /* [R,n] (n_prev > 0) */
    const float* pair_tmp;       /* spf_geo_forward's per-pair scratch of the n - n_prev new samples per ray */
    const int32_t* pair_off;
    const int32_t* slot_point;   /* [R * (n - n_prev)] */
    float* sdf_cur;              /* [R,n] written by phases 0 / 2, read by phase 1 */
    float* beta;                 /* [R] in (it > 0) / out */
    const float* beta0;          /* device scalar */
    float eps, bound_coef, add_tiny;
    int32_t beta_iters;
    const float* u_more;
    int32_t N_more;
    const float* u_fin;
    int32_t N_fin;
    float* z_merged;
    int32_t* merged_out;
    float* points_new;
    const int32_t* sel;          /* [Ne] extra samples of the current z (:541-547) */
    int32_t Ne;
    float near, far;
    const float* cam_loc;
    const float* ray_dirs;
    float* z_out;
    float* points_out;
    int32_t SR;
    int32_t* slot_sample;
    uint8_t* ray_valid;
    int32_t* flags;              /* [max_total_iters + 2], flags[0] = 1, the rest zero before the first iteration */
    int32_t it;
} spf_sampler_eval_args;
int spf_sampler_eval(const spf_sampler_eval_args* args, const spf_grid* grid, int32_t R, int32_t phase, void* stream);

/* ABI 5 — the optimisation step's sampler pass (ErrorBoundSampler_pn.get_z_vals with fast = 1: one iteration, final sampling; ray_sampler.py:
 * 377-574) behind the SDF kernel as ONE launch, wave per ray: (i) the per-sample SDF from the geometry kernel's per-pair scratch (spf_geo_forward
 * with sdf == NULL: pair_tmp, pair_off, slot_point [R*n] = point id of a sample or -1; 1000 where a sample has no neighbour), (ii) beta by
 * bisection and the N samples by inverse-CDF (spf_sampler_iter's final pass; u [R,N] per-ray uniforms), (iii) spf_sampler_finish (z_out
 * [R, N + 2 + Ne] sorted, points [R, N + 2 + Ne, 3]) and (iv) the slot assignment of the main pass's kNN (spf_grid_query's first stage):
 * slot_sample [R,SR] = the ray's first SR samples inside the grid's dilated occupancy, ray_valid [R] cleared (-> spf_grid_knn).
 * Same values as the four separate launches. */
int spf_sampler_train(const float* z, const float* pair_tmp, const int32_t* pair_off, const int32_t* slot_point, const float* beta0,
                      int32_t R, int32_t n, float eps, float bound_coef, int32_t beta_iters, const float* u, int32_t N,
                      const int32_t* sel, int32_t Ne, float near, float far, const float* cam_loc, const float* ray_dirs,
                      const spf_grid* grid, int32_t SR, float* beta_out, float* z_out, float* points, int32_t* slot_sample,
                      uint8_t* ray_valid, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-ray compositing — replaces filter_points (spurfies/model/pointneus_disent.py:207-239),
 * LaplaceDensity (spurfies/model/density.py:16-30), volume_rendering (:894-908) and the composites
 * (:765-795), forward and backward.  All arrays are dense [R,SR] (invalid slots masked).
 * ---------------------------------------------------------------------------------------- */

/* loc [R,SR,3] slot sample positions (spf_grid_query), slot_valid [R,SR], cam_loc / ray_dirs [R,3] ->
 *   z [R,SR]       t = nanmean_xyz((loc - o)/d) at valid slots, 0 elsewhere
 *   deltas [R,SR]  max(z[s+1] - z[s], 0) with z[SR] := 0; 0 at invalid slots
 *   x [R*SR,3]     o + z d  (the shading points handed to the MLP kernels) */
int spf_filter_points(const float* loc, const uint8_t* slot_valid, const float* cam_loc, const float* ray_dirs,
                      int32_t R, int32_t SR, float* z, float* deltas, float* x, void* stream);

/* sigma = Laplace(sdf; *beta) at valid slots, E = delta sigma, w = (1 - e^-E) exp(-sum_{i<j} E_i);
 * rgb = sum w c, depth = sum w z / (sum w + 1e-8), dist = sum w z / (sum w + 1e-10), acc = sum w.
 * colors [R,SR,3] must be 0 at invalid slots; beta is a DEVICE scalar (|beta_param| + beta_min).
 * pts_rendered [R,3] (may be NULL; needs cam_loc / ray_dirs [R,3]): the rendered surface points cam_loc + ray_dirs * dist the
 * pseudo-point loss queries (pointneus_disent.py:765-767), formed here instead of by a separate elementwise launch.
 * ABI 6 — evaluation outputs from the same launch (all may be NULL): grad [R,SR,3] (d sdf/d x of the slots) -> normal [R,3] =
 * sum_j w_j grad_j / |grad_j| over the valid slots (the `normal_map` of pointneus_disent.py:797-815, 886-888); ray_valid [R]: depth[r] =
 * depth_fill for a ray without a valid slot (the reference initialises `depth_values` with ones, :822-826). */
int spf_render_forward(const float* sdf, const uint8_t* slot_valid, const float* z, const float* deltas,
                       const float* colors, const float* beta, int32_t R, int32_t SR, float* weights,
                       float* rgb, float* depth, float* dist, float* acc, const float* cam_loc, const float* ray_dirs,
                       float* pts_rendered, const float* grad, float* normal, const uint8_t* ray_valid, float depth_fill, void* stream);

/* ABI 5 — the colour composite on its own.  spf_render_forward with colors == rgb == NULL is the WEIGHTS-ONLY form: everything that depends
 * on the SDF alone (weights, depth, dist, acc, pts_rendered — and with them the whole pseudo-point pass, pointneus_disent.py:765-780) can then
 * be issued right behind the geometry kernel, beside the colour MLPs instead of behind them (a forked hipGraph branch of the optimisation
 * step); spf_render_rgb then forms rgb [R,3] = sum_j weights[r,j] colors[r,j,:] with the fused kernel's lane assignment and reduction
 * tree (same bits).  spf_render_rgb_backward: g_colors [R,SR,3] = weights g_rgb and g_weights [R,SR] = colors . g_rgb — the latter is what
 * spf_render_backward takes as g_weights when it is called with g_rgb == colors == g_colors == NULL (the same sum as the fused form). */
int spf_render_rgb(const float* weights, const float* colors, int32_t R, int32_t SR, float* rgb, void* stream);
int spf_render_rgb_backward(const float* weights, const float* colors, const float* g_rgb, int32_t R, int32_t SR, float* g_colors,
                            float* g_weights, void* stream);

/* Gradients of a scalar loss given g_weights [R,SR] (may be NULL), g_rgb [R,3], g_depth [R] (may be
 * NULL), g_dist [R] (may be NULL): g_sdf [R,SR], g_colors [R,SR,3], and g_beta[0] += dL/d beta — or, when the raw
 * LaplaceDensity parameter is passed in beta_param (beta = |beta_param| + beta_min, spurfies/model/density.py:28-30),
 * g_beta[0] += sign(beta_param) dL/d beta, i.e. the parameter's own gradient.
 * g_acc [R] (may be NULL): gradient of acc = sum_j w_j, added to every slot's g_weights; g_pts_rendered [R,3] (may be NULL; needs
 * ray_dirs): gradient of spf_render_forward's pts_rendered, entering through dist (g_dist[r] += g_pts_rendered[r] . ray_dirs[r]).
 * g_beta_fixed (may be NULL): a one-entry fixed-point accumulator (see "Reproducible gradients" below) that receives the rays' terms
 * instead of g_beta's float atomics.
 * ABI 6 — lfirst [R], lcoef [R,2] (spf_local_forward) and lscale[0] (spf_loss_backward*), all three or none: the feature-consistency
 * term's gradient joins here, g_sdf[r, lfirst[r] + i] += lscale[0] * lcoef[r, i] for i in {0, 1} where lfirst[r] >= 0 (its backward has
 * no launch of its own in the optimisation step). */
int spf_render_backward(const float* sdf, const uint8_t* slot_valid, const float* z, const float* deltas,
                        const float* colors, const float* beta, const float* weights, const float* g_weights,
                        const float* g_rgb, const float* g_depth, const float* g_dist, int32_t R, int32_t SR,
                        float* g_sdf, float* g_colors, float* g_beta, const float* beta_param, const float* g_acc,
                        const float* g_pts_rendered, const float* ray_dirs, int64_t* g_beta_fixed, const int32_t* lfirst,
                        const float* lcoef, const float* lscale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight-gradient GEMM with a device-side row count — replaces autograd's AddmmBackward GEMMs for the
 * nn.Linear layers of F_color / R (spurfies/model/pointneus_disent.py:76-107).
 *   dW[256, 0:C] (leading dimension ldw) += G[0:rows, 256]^T * A[0:rows, 0:C] (leading dimension lda),
 * rows = min(*n_rows, max_rows) read on the device (NULL: max_rows).  C in {256 | multiple of 4 <= 128 | <= 32};
 * workspace: spf_wgrad_workspace_floats(C) floats (per-workgroup partial slabs, summed in a fixed order). */
int64_t spf_wgrad_workspace_floats(int32_t C);
/* arith (see SPF_ARITH_*) applies for C > 32; narrower operands always take the fp32-MFMA kernel.
 * layout: 0 = both operands are row-major [rows, .]; SPF_WGRAD_G_TILES / SPF_WGRAD_A_TILES (or-ed): that operand is stored as
 * K-MAJOR BLOCKS of 16 rows x 256 features, element (row, feature f) at ((256 (row / 16) + f) 16 + row % 16) — what
 * spf_color_forward (act1, act2) writes with SPF_ARITH_SPLIT: one 16-row MFMA K-step of an
 * operand is a contiguous 16-KB run with each feature's rows adjacent.  Blocked operands need SPF_ARITH_SPLIT, C > 32,
 * max_rows % 16 == 0 (whole blocks allocated) and, for A, C = 256; lda is ignored for a blocked A.
 * SPF_WGRAD_G_TILES64: G in K-major TILES of 64 rows, element (row, f) at ((256 (row / 64) + f) 64 + row % 64): what
 * spf_color_backward writes for G3 / G2 / G1 (full 128-byte lines from its registers); max_rows % 64 == 0. */
#define SPF_WGRAD_G_TILES 1
#define SPF_WGRAD_A_TILES 2
#define SPF_WGRAD_G_TILES64 4
/* or-ed into `layout` (spf_wgrad) / passed as `flags` (spf_wgrad_batched): the workgroups' partial slabs are summed in block order by ONE
 * reduce slice and added onto dW with a plain add, the bias gradient likewise from per-workgroup column sums — bit-reproducible run to run
 * (the default sums 16 slices with float atomics; ~4x the reduce time, < 0.3 ms per step).  SPF_ARITH_SPLIT for C > 32. */
#define SPF_WGRAD_DETERMINISTIC 8

/* dbias (may be NULL): float[256], dbias[o] += sum_rows G[row][o] — the bias gradient of the same layer, taken on the way
 * (free in the default arithmetic; a separate pass over G otherwise).
 * col_rot / col_mod (0 / 0 = off; SPF_ARITH_SPLIT, C > 32): product column i < col_mod is accumulated into dW column
 * (i + col_rot) mod col_mod, columns i >= col_mod are dropped (ldw >= col_mod).  F_color.0's input is held as [latent 64 | offset 3 |
 * encoding 36 | pad] by the colour kernels and as [offset + encoding 39 | latent 64] by the reference (pointneus_disent.py:327-331): with
 * col_rot 39, col_mod 103 the 104-wide product lands in the reference's column order without a permutation pass. */
int spf_wgrad(const float* G, const float* A, int32_t lda, int32_t C, const int32_t* n_rows, int32_t max_rows,
              float* dW, int32_t ldw, float* dbias, float* workspace, int32_t layout, int32_t arith, int32_t col_rot,
              int32_t col_mod, void* stream);

/* Up to SEVEN weight-gradient GEMMs (three before ABI 5) in one pair of launches, side by side on the chip (each problem gets a share of the
 * workgroups proportional to its work): dW_q += G_q^T A_q, dbias_q += column sums of G_q (may be NULL).  The problems share the call's row
 * count (n_rows / max_rows) unless a problem names its own (ABI 5: spf_wgrad_problem.max_rows > 0, with its n_rows): the head stage's GEMMs
 * (K = valid points) then ride in the colour trunk's launch (K = pairs) — one pipeline ramp, tail and slab reduce per step.
 * Launched one after the other each GEMM pays the pipeline ramp, a tail on a mostly idle chip and a dispatch gap — the head stage's three
 * (K = valid points) and, since ABI 4, the colour trunk's three (K = pairs).  Per problem (C, layout) must be one of
 *   (256, 0)                                         both operands row-major                      (C == 0 means 256)
 *   (256, SPF_WGRAD_G_TILES64 | SPF_WGRAD_A_TILES)   what spf_color_backward / spf_color_forward write for layers 2 and 4
 *   (36..128 step 4, SPF_WGRAD_G_TILES64)            the trunk's first layer (C = 104), A row-major with leading dimension lda
 *   (4..128 step 4, 0)                               ABI 5: both row-major, narrow A (R.0's view-encoding columns: C = 24, col_mod = 21)
 * with spf_wgrad's col_rot / col_mod per problem; anything else: spf_wgrad.  `problems` is a HOST array; workspace:
 * n_problems * spf_wgrad_workspace_floats(256) floats; tiled operands need (their) max_rows % 64 == 0. */
struct spf_wgrad_problem {
    const float* G;
    const float* A;
    int32_t lda;
    float* dW;
    int32_t ldw;
    float* dbias;
    int32_t C;          /* ABI 4 */
    int32_t layout;
    int32_t col_rot;
    int32_t col_mod;
    const int32_t* n_rows;   /* ABI 5: this problem's own row count (device, may be NULL = max_rows), used when max_rows > 0 */
    int32_t max_rows;        /* ABI 5: 0 = the call's n_rows / max_rows */
};
int spf_wgrad_batched(const struct spf_wgrad_problem* problems, int32_t n_problems, const int32_t* n_rows, int32_t max_rows,
                      float* workspace, int32_t arith, int32_t flags, void* stream);

/* ------------------------------------------------------------------------------------------
 * Latent tables
 * ---------------------------------------------------------------------------------------- */

/* dst[idx[m], :] += src[m, :] for m < M (idx < 0 skipped); c in {4, 32, 64} floats per row.
 * Backward of the reference's index_select gathers of latent rows (spurfies/model/utils.py:158-161);
 * float atomics (summation order not reproducible run to run). */
int spf_scatter_add_rows(const float* src, const int32_t* idx, int64_t m, int32_t c, float* dst, void* stream);

/* Total-variation regulariser of the geometry latents over the static neighbour graph
 * (spurfies/model/utils.py:221-282): nbr[n,k] int32 (any valid index where w == 0), w[n,k] inverse-
 * distance weights (0 = absent), norm[n] = sum_j w.  tv[i] = sum_j w_ij |f_j - f_i|_1 / norm_i
 * (the caller takes the mean).  Backward accumulates into g_feat_geo[n,32] (float atomics); the upstream gradient of tv_i is
 * g_tv[i * g_tv_stride] * scale (stride 0 + scale 1/n: the gradient of the mean, one device scalar for all points).
 * The mean over the points (`tv_regul`'s return value, utils.py:282) is the caller's: spf_loss_forward takes the tv[n] array itself. */
int spf_tv_forward(const float* feat_geo, const int32_t* nbr, const float* w, const float* norm, int32_t n,
                   int32_t k, float* tv, void* stream);
int spf_tv_backward(const float* feat_geo, const int32_t* nbr, const float* w, const float* norm,
                    const float* g_tv, int32_t g_tv_stride, float scale, int32_t n, int32_t k, float* g_feat_geo,
                    int64_t* g_feat_geo_fixed, void* stream);

/* Reproducible gradients.  The three latent-gradient scatters (spf_color_backward -> g_feat_color, spf_geo_backward_latents and
 * spf_tv_backward -> g_feat_geo) and the forward's weighted mean (spf_color_forward -> agg3, up to four partial sums per entry) add
 * with float atomics by default: the sum depends on the order the atomics land in (run-to-run
 * noise in the last bits).  With a non-NULL `*_fixed` argument (int64 [N, 64 | 32], zero before the first use) they instead add
 * every fp32 term as a 2^-48 fixed-point integer (64-bit integer atomics: associative, so order-independent; a term is exact when
 * its last mantissa bit is >= 2^-48, smaller ones round at 3.6e-15 absolute; |sum| < 2^13), and spf_fixed_accumulate rounds the sums
 * to fp32 once: dst[i] += acc[i] * 2^-48, acc[i] = 0.  This is the
 * deterministic alternative to sorting the pairs by neighbour (the reference's index_add_ is itself atomic and unordered).
 * BUFFER CONTRACT (ABI 3): every `*_fixed` pointer addresses the accumulators of an allocation that holds ONE MORE int64 directly in
 * front of them — the status word acc[-1], zero before the first use.  A non-finite term (NaN, Inf, |v| >= 2^14) adds nothing and sets
 * the status word instead; spf_fixed_accumulate then writes NaN to ALL n destination entries (so the optimiser's non-finite guard skips
 * the update, train.py:548-564, whatever the number and signs of the offending terms) and clears the word.
 * The remaining atomically summed gradients have the same option: spf_rhead_backward (R.4's weight / bias), spf_render_backward (beta) take
 * `*_fixed` accumulators, and spf_wgrad / spf_wgrad_batched sum their partial slabs and column sums in a fixed order with
 * SPF_WGRAD_DETERMINISTIC — with all of them on, one optimisation step's gradients are bit-identical from run to run. */
int spf_fixed_accumulate(int64_t* acc, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray set-up of one view
 * ---------------------------------------------------------------------------------------- */

/* uv [R,2] pixel coordinates, pose [4,4] camera-to-world (row-major), intrinsics [k_stride,k_stride] (3 or 4) ->
 *   ray_dirs [R,3]    unit world-space directions  normalize(R_cw lift(uv) + t - t)
 *   cam_loc [R,3]     the camera centre repeated per ray
 *   depth_scale [R]   z of the camera-space unit direction (the model's `depth_scale`)
 * Replaces rend_util.get_camera_params + lift (spurfies/utils/rend_util.py:60-95,143-156) as called twice per forward
 * at spurfies/model/pointneus_disent.py:640-650 (batch of one view, matrix poses).
 * beta_out (DEVICE scalar, may be NULL; needs beta_param): the forward's effective Laplace scale |*beta_param| + beta_min
 * (LaplaceDensity.get_beta, spurfies/model/density.py:28-30), which the sampler and the compositing kernels of the same forward read —
 * formed by this first launch of the forward instead of two elementwise launches. */
int spf_camera_rays(const float* uv, const float* pose, const float* intrinsics, int32_t k_stride, int32_t R,
                    float* ray_dirs, float* cam_loc, float* depth_scale, const float* beta_param, float beta_min,
                    float* beta_out, void* stream);

/* ABI 5: spf_camera_rays and spf_sampler_uniform in one launch (the first two launches of every optimisation step): the same ray arrays,
 * plus z [R,n] and points [R,n,3] of UniformSampler.get_z_vals (ray_sampler.py:33-59; t_rand [R,n] may be NULL).
 * tv_feat != NULL: spf_tv_forward (tv_feat [tv_n,32], tv_nbr / tv_w [tv_n,tv_k], tv_norm [tv_n] -> tv_out [tv_n]) rides in the same launch —
 * the TV term depends on nothing the step computes.  packs != NULL: so do spf_color_pack and spf_rhead_pack (weights as in those entry
 * points, `*_zero` = the buffer each of them clears on the way) — the "step prologue": everything that only reads parameters and the batch. */
typedef struct spf_prologue_packs {
    const float *cw0, *cb0, *cw2, *cb2, *cw4, *cb4;                 /* F_color.0 / 2 / 4 (spf_color_pack) */
    float* c_packed;
    float* c_zero;
    int64_t c_zero_floats;
    const float *rw6, *rb6, *rw0, *rb0, *rw2, *rb2, *rw4, *rb4;     /* F_color.6, R.0 / 2 / 4 (spf_rhead_pack) */
    float* r_packed;
    float* r_zero;
    int64_t r_zero_floats;
} spf_prologue_packs;
int spf_camera_uniform(const float* uv, const float* pose, const float* intrinsics, int32_t k_stride, int32_t R,
                       float* ray_dirs, float* cam_loc, float* depth_scale, const float* beta_param, float beta_min,
                       float* beta_out, const float* tlin, const float* t_rand, int32_t n, float near, float far, float* z,
                       float* points, const float* tv_feat, const int32_t* tv_nbr, const float* tv_w, const float* tv_norm,
                       int32_t tv_n, int32_t tv_k, float* tv_out, const spf_prologue_packs* packs, void* stream);

/* ------------------------------------------------------------------------------------------
 * ABI 6 — multi-view feature-consistency ("local") term of the DTU recipe (config/ours.yaml:16, local_weight 0.5) — replaces
 *   find_surface_points   spurfies/model/pointneus_disent.py:586-612
 *   the surface points    spurfies/model/pointneus_disent.py:727-749
 *   get_local_loss        spurfies/feat_utils.py:377-451 (idx_world2cam / idx_cam2img :43-55, normalize_for_grid_sample / get_in_range
 *                         :58-77, F.grid_sample(bilinear, zeros, align_corners=False), cosine term and masks :425-437)
 * and autograd's backward through them.  The per-view data arrives as a DESCRIPTOR IN DEVICE MEMORY, so that a captured hipGraph of the
 * optimisation step serves every training view: the caller overwrites the descriptor's bytes between replays (one small copy).
 * ---------------------------------------------------------------------------------------- */
#define SPF_LOCAL_MAX_VIEWS 8   /* reference view + up to 7 source views (datasets/dtu.py:268-291 passes 2) */
typedef struct spf_local_desc {
    uint64_t feat[SPF_LOCAL_MAX_VIEWS];    /* device addresses of float32 [C,H,W] feature maps: [0] = local_data["feat"], [1 + s] = feat_src[s] */
    float cam[SPF_LOCAL_MAX_VIEWS][2][16]; /* row-major 4x4 pairs {world -> camera, K in [:3,:3]}: [0] = local_data["cam"], [1 + s] = src_cams[s] */
    float center[3];                       /* local_data["center"] */
    float size;                            /* local_data["size"]:  p_world = p / 2 * size + center (feat_utils.py:405-408) */
    int32_t n_src, C, H, W;                /* source views m (1 .. SPF_LOCAL_MAX_VIEWS - 1); channels; map height / width (half the image's) */
} spf_local_desc;

/* sdf, z [R,SR]: the SDF (1000 at slots without a point) and depth of every shading slot; cam_loc, ray_dirs [R,3]; desc: DEVICE pointer.
 * Per ray: the first slot pair whose SDF goes from + to - (no crossing: lfirst = -1 and zeros), t = linearly interpolated depth of the
 * zero, surface point cam_loc + ray_dirs t, projected into the reference and the m source views, features sampled bilinearly,
 *   d_surface [R]  t (0 without a crossing)                         == find_surface_points' d_surface
 *   lfirst [R]     first slot of the crossing or -1                 (>= 0 == find_surface_points' network_mask)
 *   lsum [R]       sum over the source views of |1 - cos| where both projections are in range and |1 - cos| < 0.5
 *   lcoef [R,2]    d lsum / d sdf[r, lfirst + {0, 1}]               (the only differentiable inputs)
 * The reference's loss is sum_r lsum[r] / (m * #{lfirst >= 0}) (0 when no ray has a crossing); spf_loss_forward forms it. */
int spf_local_forward(const spf_local_desc* desc, const float* sdf, const float* z, const float* cam_loc, const float* ray_dirs, int32_t R,
                      int32_t SR, float* d_surface, int32_t* lfirst, float* lsum, float* lcoef, void* stream);

/* The backward as a dense array, for callers outside the fused optimisation step: g_sdf [R,SR] = g_sum[0] * d (sum_r lsum[r]) / d sdf
 * (zero except at the crossings; every entry is written). */
int spf_local_backward(const int32_t* lfirst, const float* lcoef, const float* g_sum, int32_t R, int32_t SR, float* g_sdf, void* stream);

/* ------------------------------------------------------------------------------------------
 * Loss terms of one optimisation step — replaces VolSDFLoss.forward (spurfies/model/loss.py:42-49,
 * 51-101) and the pseudo-point term of spurfies/model/pointneus_disent.py:765-780.
 * ---------------------------------------------------------------------------------------- */
typedef struct spf_loss_weights {
    float rgb, eikonal, tv, local, pseudo; /* config/ours.yaml:15-20 */
    int32_t world;                         /* ranks sharing the batch (tv / constant terms are split over them) */
} spf_loss_weights;

/* ABI 6: the feature-consistency term inside the loss kernels (NULL: the term is 0).  Device pointers in a host struct, read at call time. */
typedef struct spf_local_terms {
    const float* lsum;           /* [R]  spf_local_forward */
    const int32_t* lfirst;       /* [R]  spf_local_forward */
    const spf_local_desc* desc;  /* device: n_src is read from it (count = n_src * #{lfirst >= 0}) */
    float* lscale;               /* [1] OUT of the backward entry points: g_total * weights.local / count, for spf_render_backward */
} spf_local_terms;

int64_t spf_loss_workspace_floats(void);

/* rgb, rgb_gt [R,3]; acc [R] = sum_j w_j; mask_gt[r * mask_stride], r < R; grad [rows,3] + slot_valid [rows] (d sdf/dx of the shading slots;
 * NULL skips the eikonal term) with n_points[0] = number of valid slots (device); psdf [R] SDF at the rendered points with
 * pvalid / ray_valid [R] (NULL skips the pseudo term); tv (NULL: 0) = the TV term: with n_tv = 0 a device scalar (the mean itself), with
 * n_tv > 0 spf_tv_forward's per-point array tv[n_tv], whose mean (utils.py:282) is formed here; denom = NULL or device
 * {R_total, P_total, pseudo_count_total, local_count_total} (global counts of a ray-sharded batch); local = NULL or the
 * feature-consistency term's per-ray values (ABI 6).
 *   total[0]  the weighted loss;   terms[8] = {loss, rgb, eikonal, tv, mask, local, pseudo, local pseudo count}
 *   den[8]    normalisers for spf_loss_backward (five used).   workspace: spf_loss_workspace_floats() floats.
 * ABI 5: total == terms == den == NULL runs the partial-sum launch only; spf_loss_backward_finalize then forms the terms on its way (the
 * workspace must stay untouched in between). */
int spf_loss_forward(const float* rgb, const float* rgb_gt, const float* acc, const float* mask_gt, int32_t mask_stride,
                     const float* grad, const uint8_t* slot_valid, int64_t rows, const int32_t* n_points, const float* psdf,
                     const uint8_t* pvalid, const uint8_t* ray_valid, const float* tv, int32_t n_tv, const float* denom, int32_t R,
                     const spf_loss_weights* weights, float* workspace, float* total, float* terms, float* den,
                     const spf_local_terms* local, void* stream);

/* g_total = dL/d total (device scalar) -> g_rgb [R,3], g_acc [R], g_psdf [R] (may be NULL), g_tv[0] (may be NULL): the gradient w.r.t. the
 * TV mean (n_tv = 0) or w.r.t. every tv[i] of the per-point array (n_tv > 0: the mean's gradient / n_tv, one scalar for all points).
 * The eikonal term has no gradient w.r.t. any trainable tensor.  local != NULL: local->lscale[0] = g_total * weights.local /
 * count — the feature-consistency term's gradient is applied by spf_render_backward (lfirst, lcoef, lscale). */
int spf_loss_backward(const float* g_total, const float* den, const spf_loss_weights* weights, const float* rgb,
                      const float* rgb_gt, const float* acc, const float* mask_gt, int32_t mask_stride, const float* psdf,
                      const uint8_t* pvalid, const uint8_t* ray_valid, int32_t R, float* g_rgb, float* g_acc,
                      float* g_psdf, float* g_tv, int32_t n_tv, const spf_local_terms* local, void* stream);

/* ABI 5: spf_loss_backward behind a partial-sums-only spf_loss_forward — every workgroup derives the normalisers from the workspace's partial
 * sums itself and the first one also writes total / terms / den (the forward's outputs), so the step has no single-block finalize launch
 * between the partial sums and the backward (3 launches -> 2).  rows / n_points / tv / n_tv / denom: as given to spf_loss_forward (rows = 0
 * when it had no grad).  tv_g_feat != NULL (needs the per-point TV form, n_tv > 0): spf_tv_backward with the TV term's gradient rides in the
 * same launch — tv_g_feat [n_tv,32] += d loss / d tv_i * d tv_i / d latents (float atomics), graph = tv_feat, tv_nbr, tv_w, tv_norm, tv_k. */
int spf_loss_backward_finalize(const float* g_total, const spf_loss_weights* weights, const float* rgb, const float* rgb_gt,
                               const float* acc, const float* mask_gt, int32_t mask_stride, const float* psdf, const uint8_t* pvalid,
                               const uint8_t* ray_valid, int32_t R, float* g_rgb, float* g_acc, float* g_psdf, float* g_tv, int32_t n_tv,
                               const float* workspace, int64_t rows, const int32_t* n_points, const float* tv, const float* denom,
                               float* total, float* terms, float* den, const float* tv_feat, const int32_t* tv_nbr, const float* tv_w,
                               const float* tv_norm, int32_t tv_k, float* tv_g_feat, const spf_local_terms* local, void* stream);

/* ------------------------------------------------------------------------------------------
 * Parameter update — replaces the tail of the reference's train step (spurfies/train.py:359-363, 548-564):
 * clip_grad_norm_(parameters, max_norm), the NaN / Inf gradient guard, torch.optim.Adam.step().
 * ---------------------------------------------------------------------------------------- */
int64_t spf_adam_workspace_floats(void);

/* param, grad, exp_avg, exp_avg_sq: flat fp32 arrays of n elements (all trainable tensors back to back).
 *   norm = |grad|_2.  Not finite: nothing is written except state[1] += 1 (the reference then "does not update").
 *   Otherwise grad *= min(1, max_norm / (norm + 1e-6)) in place (max_norm <= 0: no clipping), t = ++state[0], and
 *   exp_avg, exp_avg_sq, param advance as torch.optim.Adam does (no weight decay, no amsgrad).
 * zero_grads != 0: grad is left ZERO instead (also when the update was skipped) — the next step's optimizer.zero_grad() (train.py:357)
 *   folded into this sweep; the clipped gradient is then not observable afterwards.
 * state: device float[4] = {t, skipped steps, last norm, last clip coefficient}, zero-initialised by the caller.
 * workspace: spf_adam_workspace_floats() floats.  Two launches, no host synchronisation. */
int spf_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, double max_norm, int32_t zero_grads, float* state, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPURFIES_HIP_H */
