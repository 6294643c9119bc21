// mesh.cpp -- MeshCreator of the host mirror: PLY / OBJ export of a PointCloudImage (file formats and numbering of
// Duke/meshcreator.cpp:16-166).
//
// What the files contain, stated as a spec:
//   * a vertex for every pixel (i, j) of the w x h cloud that holds at least one point, enumerated with the column index i
//     outermost and the row index j innermost; PLY numbers them from 0, OBJ from 1;
//   * for every pixel (i, j), in the same order, up to two triangles over its right and lower / upper-right neighbours:
//     (i,j) (i+1,j) (i,j+1) and (i,j) (i+1,j-1) (i+1,j), each only if all three corners are vertices;
//   * the reference keeps the numbers in an int image where 0 also means "no vertex", so in a PLY file the very first vertex
//     (number 0) can never be a triangle corner.  Reproduced: `usable` below.
//   * PLY: "x y z r g b" per vertex with the colour components in the order the reference prints them (c[2] c[1] c[0]); numbers
//     in default ostream formatting (%g).  OBJ: "v x y z", faces as "f a/a b/b c/c".
// The numbering is an exclusive prefix sum over the occupancy flags in column-major order: slr_prefix_index does it on the
// GPU (kernels_compact.hip); the text is assembled in memory and written once.
#include "duke.hpp"

#include <stdio.h>
#include <string.h>

#include <fstream>

namespace duke {

namespace {

constexpr uint32_t kNoVertex = 0xFFFFFFFFu;

struct Numbering {
    std::vector<uint32_t> id;            // [h][w] vertex number or kNoVertex
    uint32_t count = 0;
    bool ok = false;
};

Numbering number_vertices(const PointCloudImage &cloud, uint32_t first, slr_ctx *ctx)
{
    Numbering n;
    const int w = cloud.getWidth(), h = cloud.getHeight();
    n.id.assign((size_t)w * h, kNoVertex);
    slr_ctx *own = nullptr;
    if (!ctx) {                                          // no reconstructor context handed over: one on the current device
        int dev = 0;
        if (slr_current_device(&dev) != SLR_OK || slr_create(dev, &own) != SLR_OK) return n;   // no GPU: the caller reports it
        ctx = own;
    }
    n.ok = slr_prefix_index(ctx, cloud.numOfPointsForPixel.data(), w, h, 1, first, kNoVertex, n.id.data(), &n.count, SLR_MEM_HOST) == SLR_OK;
    if (own) slr_destroy(own);
    return n;
}

void put(std::string &s, float v) { char b[32]; s.append(b, (size_t)snprintf(b, sizeof b, "%g", (double)v)); }
void put(std::string &s, unsigned v) { char b[16]; s.append(b, (size_t)snprintf(b, sizeof b, "%u", v)); }

// calls emit(a, b, c) for every triangle, in file order; returns how many
template <typename F>
size_t triangles(const Numbering &n, int w, int h, uint32_t first, F emit)
{
    auto usable = [&](int i, int j) -> uint32_t {        // a corner's number, or kNoVertex when it cannot be one
        if (i < 0 || j < 0 || i >= w || j >= h) return kNoVertex;
        const uint32_t v = n.id[(size_t)j * w + i];
        return (v == kNoVertex || (first == 0 && v == 0)) ? kNoVertex : v;
    };
    size_t m = 0;
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            const uint32_t a = usable(i, j), right = usable(i + 1, j);
            if (a == kNoVertex || right == kNoVertex) continue;
            const uint32_t below = usable(i, j + 1), upright = usable(i + 1, j - 1);
            if (below != kNoVertex) { emit(a, right, below); m++; }
            if (upright != kNoVertex) { emit(a, upright, right); m++; }
        }
    return m;
}

}  // namespace

MeshCreator::MeshCreator(PointCloudImage *in, slr_ctx *numbering_ctx) : cloud(in), ctx(numbering_ctx), w(in->getWidth()), h(in->getHeight()) {}

bool MeshCreator::exportPlyMesh(const std::string &path)
{
    const Numbering n = number_vertices(*cloud, 0, ctx);
    if (!n.ok) return false;
    const size_t faces = triangles(n, w, h, 0, [](uint32_t, uint32_t, uint32_t) {});
    std::string s;
    s.reserve((size_t)n.count * 48 + faces * 24 + 512);
    s += "ply\nformat ascii 1.0\nelement vertex "; put(s, n.count);
    s += "\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face ";
    put(s, (unsigned)faces);
    s += "\nproperty list uchar int vertex_indices\nend_header\n";
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            Point3f p;
            int c[3];
            if (!cloud->getPoint(i, j, p, c)) continue;
            put(s, p.x); s += ' '; put(s, p.y); s += ' '; put(s, p.z);
            for (int k = 2; k >= 0; k--) { s += ' '; put(s, (unsigned)(c[k] < 0 ? 0 : c[k])); }
            s += '\n';
        }
    triangles(n, w, h, 0, [&](uint32_t a, uint32_t b, uint32_t c) {
        s += "3 "; put(s, a); s += ' '; put(s, b); s += ' '; put(s, c); s += '\n';
    });
    std::ofstream out(path.c_str(), std::ios::binary);
    out.write(s.data(), (std::streamsize)s.size());
    return (bool)out;
}

bool MeshCreator::exportObjMesh(const std::string &path)
{
    const Numbering n = number_vertices(*cloud, 1, ctx);
    if (!n.ok) return false;
    std::string s;
    s.reserve((size_t)n.count * 40 + 512);
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            Point3f p;
            if (!cloud->getPoint(i, j, p)) continue;
            s += "v "; put(s, p.x); s += ' '; put(s, p.y); s += ' '; put(s, p.z); s += '\n';
        }
    triangles(n, w, h, 1, [&](uint32_t a, uint32_t b, uint32_t c) {
        const uint32_t v[3] = {a, b, c};
        s += 'f';
        for (int k = 0; k < 3; k++) { s += ' '; put(s, v[k]); s += '/'; put(s, v[k]); }
        s += '\n';
    });
    std::ofstream out(path.c_str(), std::ios::binary);
    out.write(s.data(), (std::streamsize)s.size());
    return (bool)out;
}

}  // namespace duke
