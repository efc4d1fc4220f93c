// CPU sanitizer harness for the index arithmetic of the K1 march (test infrastructure; built by tests/test_index_sanitizers.py
// with `hipcc --offload-host-only -fsanitize=address,undefined`: GPU sanitizers are not available on the pool).
//
// It includes the LIBRARY'S OWN sources, so prepare(), the pixel map, the cell -> element arithmetic of every layout, the
// skipping pre-pass and MapWindow's index functions are the product's (MRIRT_HD functions, csrc/mrirt_device.h), not a copy.
// What is walked, against buffers of exactly the sizes the host wrappers allocate (render.py `need`, mrirt_skip_mask_words):
//   A. every element a sample's gathers can address, for every base cell sampleLinear's clamp allows (ix <= X - 2) and every
//      voxel sampleLabel's clamp allows, in LINEAR / BRICK / VG / QUAD / VGA / LABCELL — exhaustively on small grids, on the
//      corners and edges of the big ones (incl. the >= 4 GiB `wide` case), and that 32-bit byte offsets do not wrap where
//      the kernels use them;
//   B. the workgroup / lane -> pixel map of whole frames and tile shards: every store index inside the output, every pixel
//      covered exactly once;
//   C. the skipping pre-pass as the kernels run it (ballot words, three separable passes through the two byte maps), its
//      result against a brute-force Chebyshev transform, and MapWindow::lookup's window moves for random and adversarial
//      packets (every fetch inside the map, every slot inside the wave, the byte returned = the map's byte);
//   D. the host half of mrirt_render_brats_skip / _ex for the launch of VERDICT r3 #1 (n = 72, QUAD, 3 channels, 160^2,
//      both showSeg values) and random argument blocks (no device here: the launches themselves fail with a status).
// Exit status 0 = no check failed (the sanitizers abort on their own findings).
#include "../../mri-raytracer_amd/csrc/brats_march.hip"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

using namespace mrirt;

static int g_fail = 0;
static long g_checks = 0;
#define CHECK(cond, ...)                                                                   \
    do {                                                                                   \
        ++g_checks;                                                                        \
        if (!(cond)) {                                                                     \
            if (g_fail < 40) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } \
            ++g_fail;                                                                      \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// A. tap addresses
// ---------------------------------------------------------------------------------------------------------------------
struct Sizes { uint64_t linear, brick, vec4, vga, labcell, labLinear, labBrick; };      // bytes, as render.py allocates them
static Sizes sizes_of(const uint32_t d[3]) {
    Sizes s;
    const uint64_t nvox = (uint64_t)d[0] * d[1] * d[2];
    s.linear = nvox * 4; s.brick = (uint64_t)mrirt_brick_elems(d) * 4; s.vec4 = (uint64_t)mrirt_vec4_elems(d) * 16;
    s.vga = (uint64_t)mrirt_vga_elems(d) * 16; s.labcell = (uint64_t)mrirt_vec4_elems(d) * 8;
    s.labLinear = nvox * 4; s.labBrick = (uint64_t)mrirt_brick_elems(d) * 4;
    return s;
}

static void check_cell(const uint32_t d[3], const Sizes& sz, const GridDims& lin, const GridDims& brk, const GridDims& v4,
                       const VgaDims& vga, uint32_t ix, uint32_t iy, uint32_t iz) {
    // VG (8 taps) and QUAD (2 taps): Taps<2>::issue / issue_async, Taps<3>
    {
        const CellOffsets k = vec4_cell(v4, ix, iy, iz);
        const uint32_t o10 = k.o + k.dx, o01 = k.o + k.dy, o11 = o10 + k.dy;
        const uint32_t e[8] = { k.o, o10, o01, o11, k.o + k.dz, o10 + k.dz, o01 + k.dz, o11 + k.dz };
        for (int i = 0; i < 8; ++i) {
            CHECK(((uint64_t)e[i] + 1) * 16 <= sz.vec4, "VG tap %d of cell (%u,%u,%u) in %ux%ux%u: element %u past %llu bytes", i, ix, iy, iz, d[0], d[1], d[2], e[i], (unsigned long long)sz.vec4);
            if (!v4.wide) CHECK((uint64_t)(uint32_t)(e[i] << 4) == (uint64_t)e[i] * 16, "32-bit byte offset wraps on a grid not marked wide");
        }
        // the element really is the voxel the tap means (x fastest inside the 2x2x2 brick)
        const uint32_t want = AddrVec4::ox(v4, ix + 1) + AddrVec4::oy(v4, iy + 1) + AddrVec4::oz(v4, iz + 1);
        CHECK(e[7] == want, "vec4_cell's far corner %u != ox+oy+oz %u", e[7], want);
        // LABCELL: one 8-byte element at the cell's own offset
        CHECK(((uint64_t)k.o + 1) * 8 <= sz.labcell, "label cell past its grid");
        if (sz.labcell < (1ull << 32)) CHECK((uint64_t)(uint32_t)(k.o << 3) == (uint64_t)k.o * 8, "label-cell byte offset wraps");
    }
    // VGA: the three copies (prepare() and mrirt_build_vec4_grid refuse the layout when a copy reaches 2^28 elements)
    bool vgaOk = true;
    for (int a = 0; a < 3; ++a) vgaOk = vgaOk && vga_copy_elems(d, a) < (1ull << 28);
    for (int a = 0; a < 3 && vgaOk; ++a) {
        const FlatAxis& f = vga.ax[a];
        const CellOffsets k = flat_cell(f, ix, iy, iz);
        const uint32_t o10 = k.o + k.dx, o01 = k.o + k.dy, o11 = o10 + k.dy;
        const uint32_t e[8] = { k.o, o10, o01, o11, k.o + k.dz, o10 + k.dz, o01 + k.dz, o11 + k.dz };
        const uint64_t copyBytes = vga_copy_elems(d, a) * 16;
        for (int i = 0; i < 8; ++i) {
            CHECK(((uint64_t)e[i] + 1) * 16 <= copyBytes, "VGA copy %d tap %d of cell (%u,%u,%u): element %u past the copy (%llu B)", a, i, ix, iy, iz, e[i], (unsigned long long)copyBytes);
            CHECK(f.baseBytes + ((uint64_t)e[i] + 1) * 16 <= sz.vga, "VGA copy %d tap past the whole grid", a);
            CHECK((uint64_t)(uint32_t)(e[i] << 4) == (uint64_t)e[i] * 16, "VGA 32-bit byte offset wraps (%ux%ux%u copy %d)", d[0], d[1], d[2], a);
        }
    }
    // LINEAR: four 8-byte pairs (Taps<0,false>, TapsScalar<0>) + the shaded neighbours; BRICK: eight words + neighbours
    {
        const uint32_t o = ix + iy * lin.sY + iz * lin.sZ;
        const uint32_t e[4] = { o, o + lin.sY, o + lin.sZ, o + lin.sY + lin.sZ };
        for (int i = 0; i < 4; ++i) CHECK(((uint64_t)e[i] + 2) * 4 <= sz.linear, "LINEAR pair %d of cell (%u,%u,%u) past the grid", i, ix, iy, iz);
        using A0 = Addr<0>;
        using A1 = Addr<1>;
        const uint32_t xs[4] = { ix > 0 ? ix - 1 : 0, ix, ix + 1, std::min(ix + 2, d[0] - 1) };
        const uint32_t ys[4] = { iy > 0 ? iy - 1 : 0, iy, iy + 1, std::min(iy + 2, d[1] - 1) };
        const uint32_t zs[4] = { iz > 0 ? iz - 1 : 0, iz, iz + 1, std::min(iz + 2, d[2] - 1) };
        for (uint32_t x : xs) for (uint32_t y : ys) for (uint32_t z : zs) {
            CHECK(((uint64_t)A0::ox(lin, x) + A0::oy(lin, y) + A0::oz(lin, z) + 1) * 4 <= sz.linear, "LINEAR tap past the grid");
            CHECK(((uint64_t)A1::ox(brk, x) + A1::oy(brk, y) + A1::oz(brk, z) + 1) * 4 <= sz.brick, "BRICK tap (%u,%u,%u) past the grid", x, y, z);
        }
    }
}

static void check_label_voxel(const uint32_t d[3], const Sizes& sz, const LabelAddr& ll, const LabelAddr& lb, uint32_t x, uint32_t y, uint32_t z) {
    CHECK(((uint64_t)ll.off(x, y, z) + 1) * 4 <= sz.labLinear, "LINEAR label (%u,%u,%u) past the grid", x, y, z);
    CHECK(ll.off(x, y, z) == x + (uint64_t)d[0] * (y + (uint64_t)d[1] * z), "LINEAR label offset is not x + X (y + Y z)");
    CHECK(((uint64_t)lb.off(x, y, z) + 1) * 4 <= sz.labBrick, "BRICK label (%u,%u,%u) past the grid", x, y, z);
}

static void grid_checks(uint32_t X, uint32_t Y, uint32_t Z, bool exhaustive) {
    const uint32_t d[3] = { X, Y, Z };
    const Sizes sz = sizes_of(d);
    GridDims lin, brk, v4;
    fill_grid_dims(lin, d, MRIRT_LAYOUT_LINEAR); fill_grid_dims(brk, d, MRIRT_LAYOUT_BRICK); fill_grid_dims(v4, d, MRIRT_LAYOUT_VG);
    VgaDims vga;
    fill_vga_dims(vga, d);
    LabelAddr ll, lb;
    fill_label_addr(ll, d, MRIRT_LAYOUT_LINEAR); fill_label_addr(lb, d, MRIRT_LAYOUT_BRICK);
    CHECK((v4.wide != 0) == (sz.vec4 >= (1ull << 32)), "GridDims::wide disagrees with the grid's size");
    auto edge = [](uint32_t n) {                          // the base cells / voxels worth visiting on one axis of a big grid
        std::vector<uint32_t> v;
        for (uint32_t c : { 0u, 1u, 2u, 3u, 4u, 7u, 8u, n / 2 - 1, n / 2, n / 2 + 1, n - 9, n - 8, n - 5, n - 4, n - 3, n - 2, n - 1 })
            if (c < n) v.push_back(c);
        std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
        return v;
    };
    std::vector<uint32_t> ax[3];
    for (int k = 0; k < 3; ++k) {
        if (exhaustive) { ax[k].resize(d[k]); for (uint32_t i = 0; i < d[k]; ++i) ax[k][i] = i; }
        else ax[k] = edge(d[k]);
    }
    for (uint32_t z : ax[2]) for (uint32_t y : ax[1]) for (uint32_t x : ax[0]) {
        check_label_voxel(d, sz, ll, lb, x, y, z);
        if (x + 1 < X && y + 1 < Y && z + 1 < Z) check_cell(d, sz, lin, brk, v4, vga, x, y, z);     // sampleLinear's clamp: base <= dim - 2
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// B. pixel map (uses prepare(), i.e. the choices the library makes for a launch)
// ---------------------------------------------------------------------------------------------------------------------
static MrirtBratsParams scene(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t w, uint32_t h, uint32_t steps, uint32_t channels) {
    MrirtBratsParams p;
    memset(&p, 0, sizeof p);
    p.imageSize[0] = w; p.imageSize[1] = h; p.fovY = 0.8f;
    p.eye[0] = 0.3f; p.eye[1] = -0.2f; p.eye[2] = -2.4f;
    p.U[0] = 1.0f; p.V[1] = 1.0f; p.W[2] = 1.0f;
    const uint32_t d[3] = { n0, n1, n2 };
    const uint32_t m = std::max(n0, std::max(n1, n2));
    for (int k = 0; k < 3; ++k) { p.dims[k] = d[k]; p.voxelSize[k] = 2.0f / (float)m; p.volMin[k] = -(float)d[k] / (float)m; }
    p.stepSize = 3.5f / (float)steps;
    for (uint32_t c = 0; c < 4; ++c) { p.volEnabled[c] = c < channels; p.volWeight[c] = 1.0f; }
    p.ww = 0.7f; p.wl = 0.45f; p.intensityAlpha = 6.0f; p.gamma = 1.0f;
    for (int i = 0; i < 8; ++i) { p.lutColorAlpha[i][0] = 0.1f * i; p.lutColorAlpha[i][3] = 0.9f; }
    return p;
}

static void pixel_map_checks(const MrirtBratsParams& p, const MrirtRenderExt* ext, int64_t pitch) {
    const void* vol[4] = { (void*)0x1000, (void*)0x1000, (void*)0x1000, (void*)0x1000 };
    K1Args a;
    Prepared cfg;
    const int rc = prepare(&p, ext, vol, (void*)0x1000, (void*)0x1000, true, pitch, a, cfg);
    CHECK(rc == MRIRT_OK, "prepare() = %d", rc);
    if (rc != MRIRT_OK) return;
    const PixelMap& m = a.map;
    const uint32_t threads = m.blockPx == 8 ? 64u : 256u, grid = m.chunk * kXcds;
    const bool tiles = m.tileSize != 0;
    const int64_t outPixels = tiles ? (int64_t)mrirt_tiles_for_rank(m.width, m.height, m.tileSize, m.tileRank, m.tileWorld) * m.tileSize * m.tileSize
                                    : (int64_t)pitch * m.height;
    std::vector<uint8_t> seen((size_t)outPixels, 0);      // every store index at most once
    std::vector<uint8_t> covered((size_t)m.width * m.height, 0);
    for (uint32_t b = 0; b < grid; ++b)
        for (uint32_t t = 0; t < threads; ++t) {
            uint32_t px = 0, py = 0;
            int64_t o = -1;
            const int kind = map_pixel_at(m, b, t, px, py, o);
            if (kind == 0) continue;
            CHECK(o >= 0 && o < outPixels, "store index %lld outside the output of %lld pixels (block %u thread %u)", (long long)o, (long long)outPixels, b, t);
            if (o < 0 || o >= outPixels) continue;
            CHECK(seen[(size_t)o] == 0, "store index %lld written twice", (long long)o);
            seen[(size_t)o] = 1;
            if (kind == 1) {
                CHECK(px < m.width && py < m.height, "marching pixel (%u,%u) outside the image", px, py);
                if (px < m.width && py < m.height) { CHECK(covered[(size_t)py * m.width + px] == 0, "pixel marched twice"); covered[(size_t)py * m.width + px] = 1; }
            }
        }
    if (!tiles) {
        size_t n = 0;
        for (uint8_t c : covered) n += c;
        CHECK(n == covered.size(), "%zu of %zu pixels covered", n, covered.size());
    } else {
        for (size_t i = 0; i < seen.size(); ++i) if (!seen[i]) { CHECK(false, "compact tile slot %zu never written", i); break; }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// C. the skipping pre-pass and MapWindow, as the kernels index them
// ---------------------------------------------------------------------------------------------------------------------
static void skip_checks(uint32_t X, uint32_t Y, uint32_t Z, uint32_t seed) {
    const uint32_t d[3] = { X, Y, Z };
    const uint32_t mx = (X + 7) / 8, my = (Y + 7) / 8, mz = (Z + 7) / 8, cells = mx * my * mz;
    CHECK((int64_t)cells == mrirt_macro_cells(d), "macro cell count");
    const int64_t words = mrirt_skip_mask_words(d);
    // exactly the scratch render.py allocates: torch.empty(words, int32)
    std::vector<uint32_t> scratch((size_t)words, 0xdeadbeefu);
    uint32_t* mask = scratch.data();
    std::mt19937 rng(seed);
    // "empty" flags: a ball of tissue in air plus salt (the flags' values do not matter to the index checks)
    std::vector<uint8_t> empty(cells);
    for (uint32_t c = 0; c < cells; ++c) {
        const float x = (float)(c % mx) / mx - 0.5f, y = (float)((c / mx) % my) / my - 0.45f, z = (float)(c / (mx * my)) / mz - 0.55f;
        empty[c] = (x * x + y * y + z * z > 0.09f) && (rng() % 61u != 0u);
    }
    // skip_mask_kernel: grid of ceil(cells / 256) blocks of 256 threads; lane 0 of each wave stores the wave's ballot
    const uint32_t blocks = (cells + 255) / 256;
    for (uint32_t b = 0; b < blocks; ++b)
        for (uint32_t w = 0; w < 4; ++w) {
            const uint32_t cell0 = b * 256 + w * 64;
            uint64_t bits = 0;
            for (uint32_t l = 0; l < 64; ++l) if (cell0 + l < cells && empty[cell0 + l]) bits |= 1ull << l;
            const int64_t word = skip_ballot_word(cell0, cells);
            if (word >= 0) {
                CHECK(word + 1 < (int64_t)skip_bit_words(cells), "ballot words %lld,%lld outside the %u bit words", (long long)word, (long long)word + 1, skip_bit_words(cells));
                CHECK(word + 1 < words, "ballot word outside the scratch");
                mask[word] = (uint32_t)bits; mask[word + 1] = (uint32_t)(bits >> 32);
            } else {
                CHECK(cell0 >= cells, "a wave that holds cells stores no ballot");
            }
        }
    uint8_t* mapA = reinterpret_cast<uint8_t*>(mask + skip_bit_words(cells));
    uint8_t* mapB = mapA + skip_map_stride(cells);
    CHECK((mapB + cells) <= reinterpret_cast<uint8_t*>(mask + words), "byte map B (%u cells) runs past the scratch of %lld words", cells, (long long)words);
    CHECK((mapA + cells) <= mapB, "byte map A overlaps B");
    // skip_dist_kernel<0,1,2>: one thread per cell (threads past `cells` return)
    auto bounded = [&](const uint8_t* base, uint32_t cell) -> uint32_t {
        CHECK(cell < cells, "distance pass reads cell %u of %u", cell, cells);
        return cell < cells ? base[cell] : 0u;
    };
    for (uint32_t c = 0; c < blocks * 256; ++c) {
        if (c >= cells) continue;
        mapA[c] = (uint8_t)skip_dist_cell<0>(c, mx, my, mz, [&](uint32_t cell) -> uint32_t {
            CHECK((cell >> 5) < skip_bit_words(cells), "bit read outside the bit words");
            return skip_bit_value(mask, cell);
        });
    }
    for (uint32_t c = 0; c < cells; ++c) mapB[c] = (uint8_t)skip_dist_cell<1>(c, mx, my, mz, [&](uint32_t cell) { return bounded(mapA, cell); });
    for (uint32_t c = 0; c < cells; ++c) mapA[c] = (uint8_t)skip_dist_cell<2>(c, mx, my, mz, [&](uint32_t cell) { return bounded(mapB, cell); });
    // against the definition: r = largest r <= 31 such that every in-grid cell within Chebyshev distance r - 1 is flagged
    uint32_t leapable = 0;
    for (uint32_t c = 0; c < cells; c += (cells > 6000 ? 7 : 1)) {
        const int cx = c % mx, cy = (c / mx) % my, cz = c / (mx * my);
        uint32_t r = 0;
        for (uint32_t t = 1; t <= kSkipDistCap; ++t) {
            bool all = true;
            const int k = (int)t - 1;
            for (int z = cz - k; z <= cz + k && all; ++z) for (int y = cy - k; y <= cy + k && all; ++y) for (int x = cx - k; x <= cx + k && all; ++x)
                if (x >= 0 && y >= 0 && z >= 0 && x < (int)mx && y < (int)my && z < (int)mz && !empty[x + mx * (y + my * z)]) all = false;
            if (!all) break;
            r = t;
        }
        CHECK(mapA[c] == r, "empty radius of cell %u: %u, brute force %u", c, mapA[c], r);
        leapable += r >= 2;
    }
    CHECK(leapable > 0, "the scene has no leapable cell: the check is not doing its job");
    // MapWindow::lookup for packets of 64 samples: the window's moves and reads
    struct Win { uint32_t bytes[64]; uint32_t ox, oy, oz; } win;
    for (int trial = 0; trial < 4000; ++trial) {
        if (trial % 50 == 0) { for (auto& b : win.bytes) b = 0; win.ox = win.oy = win.oz = 0x40000000u; }     // reset()
        uint32_t cx[64], cy[64], cz[64];
        bool alive[64];
        // a packet: 64 samples clustered around a point (spread 0..6 macro cells: wider than the window now and then)
        const uint32_t bx = (uint32_t)(rng() % mx), by = (uint32_t)(rng() % my), bz = (uint32_t)(rng() % mz), spread = (uint32_t)(rng() % 7u);
        bool any = false;
        for (int l = 0; l < 64; ++l) {
            cx[l] = std::min(bx + (spread ? (uint32_t)(rng() % (spread + 1)) : 0u), mx - 1);
            cy[l] = std::min(by + (spread ? (uint32_t)(rng() % (spread + 1)) : 0u), my - 1);
            cz[l] = std::min(bz + (spread ? (uint32_t)(rng() % (spread + 1)) : 0u), mz - 1);
            alive[l] = rng() % 5u != 0u;
            any = any || alive[l];
        }
        bool in[64], move = false;
        for (int l = 0; l < 64; ++l) { in[l] = window_holds(cx[l], cy[l], cz[l], win.ox, win.oy, win.oz); move = move || (alive[l] && !in[l]); }
        if (move) {
            const bool px = rng() & 1u, py = rng() & 1u, pz = rng() & 1u;
            uint32_t lo[3] = { 255u, 255u, 255u }, hi[3] = { 0u, 0u, 0u };       // wave_min8(alive ? c : 255), wave_max8(alive ? c : 0)
            for (int l = 0; l < 64; ++l) if (alive[l]) {
                lo[0] = std::min(lo[0], cx[l]); lo[1] = std::min(lo[1], cy[l]); lo[2] = std::min(lo[2], cz[l]);
                hi[0] = std::max(hi[0], cx[l]); hi[1] = std::max(hi[1], cy[l]); hi[2] = std::max(hi[2], cz[l]);
            }
            CHECK(any, "a move without a live lane");
            win.ox = window_origin(px, lo[0], hi[0]); win.oy = window_origin(py, lo[1], hi[1]); win.oz = window_origin(pz, lo[2], hi[2]);
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t idx = window_fetch_index(win.ox, win.oy, win.oz, l, mx, my, mz, mx * my);
                CHECK(idx < cells, "window fetch %u outside the map of %u cells (origin %u,%u,%u lane %u)", idx, cells, win.ox, win.oy, win.oz, l);
                win.bytes[l] = idx < cells ? mapA[idx] : 0u;
            }
            for (int l = 0; l < 64; ++l) in[l] = window_holds(cx[l], cy[l], cz[l], win.ox, win.oy, win.oz);
        }
        for (int l = 0; l < 64; ++l) {
            const uint32_t slot = window_slot(cx[l], cy[l], cz[l], win.ox, win.oy, win.oz);
            CHECK(slot < 64, "ds_bpermute slot %u", slot);
            if (in[l]) {
                const uint32_t want = mapA[cx[l] + mx * (cy[l] + my * cz[l])];
                CHECK(win.bytes[slot] == want, "window returns %u for cell (%u,%u,%u), the map holds %u", win.bytes[slot], cx[l], cy[l], cz[l], want);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// D. host halves under the sanitizers
// ---------------------------------------------------------------------------------------------------------------------
static void host_half_checks() {
    const void* vol[4] = { (void*)0x10000, (void*)0x20000, (void*)0x30000, nullptr };
    for (uint32_t showSeg = 0; showSeg < 2; ++showSeg) {
        MrirtBratsParams p = scene(72, 72, 72, 160, 160, 192, 3);
        p.showSeg = showSeg;
        MrirtRenderExt e;
        memset(&e, 0, sizeof e);
        e.layout = MRIRT_LAYOUT_QUAD;
        pixel_map_checks(p, &e, 160);
        const uint32_t d[3] = { 72, 72, 72 };
        std::vector<uint32_t> fake((size_t)mrirt_skip_mask_words(d));
        MrirtSkip s;
        memset(&s, 0, sizeof s);
        s.macroUb[0] = s.macroUb[1] = s.macroUb[2] = (const float*)0x40000;
        s.macroSeg = (const uint32_t*)0x50000;
        s.mask = fake.data(); s.maskWords = (uint32_t)fake.size();
        const int app = mrirt_brats_skip_applicable(&p, &e, vol, (void*)0x60000, nullptr, &s);
        CHECK(app == 1, "the launch of VERDICT r3 #1 (QUAD, 3 channels, showSeg %u) is a skipping launch: got %d", showSeg, app);
        // no device in this build: the launch fails, with a status, after the whole host half has run
        const int rc = mrirt_render_brats_skip(&p, &e, vol, (void*)0x60000, nullptr, &s, (void*)0x70000, 160, nullptr, nullptr);
        CHECK(rc == MRIRT_ERR_LAUNCH || rc == MRIRT_OK, "mrirt_render_brats_skip on a machine without a GPU: %d", rc);
        s.maskWords -= 1;
        CHECK(mrirt_render_brats_skip(&p, &e, vol, (void*)0x60000, nullptr, &s, (void*)0x70000, 160, nullptr, nullptr) == MRIRT_ERR_ARG, "a short scratch must be refused");
        const int rc2 = mrirt_render_brats_ex(&p, &e, vol, (void*)0x60000, nullptr, (void*)0x70000, 160, nullptr, nullptr);
        CHECK(rc2 == MRIRT_ERR_LAUNCH || rc2 == MRIRT_OK, "mrirt_render_brats_ex: %d", rc2);
    }
    // random argument blocks through prepare(): whatever it accepts must give a consistent pixel map (UBSan: no float -> int
    // overflow, no shift past the width, no signed wrap on the way)
    std::mt19937 rng(99);
    int accepted = 0;
    for (int t = 0; t < 400; ++t) {
        const uint32_t lay = rng() % 5u;
        MrirtBratsParams p = scene(2 + rng() % 300, 2 + rng() % 300, 2 + rng() % 300, 1 + rng() % 700, 1 + rng() % 500, 1 + rng() % 400, 1 + rng() % 4);
        MrirtRenderExt e;
        memset(&e, 0, sizeof e);
        e.layout = lay;
        e.shadeMode = lay != MRIRT_LAYOUT_QUAD && (rng() & 1u);
        e.kernelVariant = (rng() & 1u) ? (rng() & 0x33fu) : 0u;
        if (rng() % 3u == 0u) { e.tileSize = 16u << (rng() % 3u); e.tileWorld = 1 + rng() % 8u; e.tileRank = rng() % e.tileWorld; }
        if (rng() % 11u == 0u) p.stepSize = (rng() & 1u) ? 0.0f : 1e-12f;                 // must be refused, not marched
        K1Args a;
        Prepared cfg;
        const int64_t pitch = p.imageSize[0] + rng() % 9u;
        const void* v4[4] = { (void*)0x1000, (void*)0x1000, (void*)0x1000, (void*)0x1000 };
        const int rc = prepare(&p, &e, v4, (void*)0x1000, (void*)0x1000, true, pitch, a, cfg);
        if (rc != MRIRT_OK) continue;
        ++accepted;
        if (t % 8 == 0) pixel_map_checks(p, &e, pitch);
    }
    CHECK(accepted > 200, "only %d of 400 random argument blocks were accepted", accepted);
}

int main() {
    // A: small grids exhaustively (odd sides, sides that are not multiples of the brick or of 8, the failing test's 72^3) ...
    const uint32_t small[][3] = { { 72, 72, 72 }, { 2, 2, 2 }, { 3, 2, 5 }, { 5, 7, 9 }, { 64, 33, 17 }, { 9, 8, 7 }, { 40, 36, 32 }, { 200, 168, 136 } };
    for (auto& s : small) grid_checks(s[0], s[1], s[2], (uint64_t)s[0] * s[1] * s[2] <= 400000);
    // ... the configurations of BASELINE.json and the >= 4 GiB fall-backs on their corners and edges
    grid_checks(256, 256, 256, false);
    grid_checks(512, 512, 512, false);
    grid_checks(645, 645, 645, false);        // a VG / QUAD grid >= 4 GiB: `wide`
    grid_checks(1024, 1024, 272, false);      // VERDICT r3 #6's wide grid
    grid_checks(1024, 1024, 1024, false);     // label grid of 2^30 voxels
    // C
    skip_checks(72, 72, 72, 1);
    skip_checks(200, 168, 136, 2);
    skip_checks(9, 8, 7, 3);                  // two macro cells per axis: every window clamps
    skip_checks(512, 512, 512, 4);
    skip_checks(2048, 16, 16, 5);             // 256 macro cells on one axis: the 8-bit reductions' limit
    // B + D
    host_half_checks();
    {
        MrirtBratsParams p = scene(64, 64, 64, 2048, 2048, 512, 1);
        MrirtRenderExt e;
        memset(&e, 0, sizeof e);
        e.layout = MRIRT_LAYOUT_VGA; e.shadeMode = 1;
        for (uint32_t world : { 1u, 2u, 3u, 8u })
            for (uint32_t ts : { 64u, 32u })
                for (uint32_t skew : { 0u, 7u, 1u, 31u, 1000003u }) {
                    e.tileSize = ts; e.tileWorld = world; e.tileRank = world - 1; e.tileSkew = skew;
                    pixel_map_checks(p, &e, 2048);
                }
        // the skewed deal is a permutation of the tile grid, and tile_dealt_index inverts tile_position (the de-tiling kernel's side)
        for (uint32_t tilesX : { 1u, 3u, 32u, 33u })
            for (uint32_t skewAsked : { 0u, 1u, 7u, 31u, 4000000000u }) {
                const uint32_t skew = skewAsked % tilesX;                       // (what fill_pixel_map / mrirt_detile hand the kernels)
                std::vector<uint8_t> hit(tilesX * 5u, 0);
                for (uint32_t t = 0; t < tilesX * 5u; ++t) {
                    uint32_t tx, ty;
                    tile_position(t, tilesX, skew, tx, ty);
                    CHECK(tx < tilesX && ty < 5u, "tile %u lands outside the grid", t);
                    if (tx < tilesX && ty < 5u) { CHECK(!hit[ty * tilesX + tx], "two tiles on one position"); hit[ty * tilesX + tx] = 1; }
                    CHECK(tile_dealt_index(tx, ty, tilesX, skew) == t, "tile_dealt_index does not invert tile_position (t %u skew %u tilesX %u)", t, skew, tilesX);
                }
            }
        e.tileSize = 0; e.tileSkew = 0; e.kernelVariant = 0;
        p.imageSize[0] = 1024; p.imageSize[1] = 1024;
        pixel_map_checks(p, &e, 1024);
        p.imageSize[0] = 203; p.imageSize[1] = 151;
        pixel_map_checks(p, &e, 211);
    }
    printf("index_harness: %ld checks, %d failed\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
