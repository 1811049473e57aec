/* integration/large_scenes/gen_mesh.c — deterministic stand-ins for the big meshes the reference tree lists in
 * .MISSING_LARGE_BLOBS. Only +,-,*,/ and sqrt on doubles and an integer
 * hash: every machine writes the same bytes (the build compiles this with -O2 -ffp-contract=off).
 *
 *   gen_mesh water <cells> <out.obj>
 *       heightfield over x,z in [-1,1]^2, (cells+1)^2 vertices, 2*cells^2 triangles (cells = 1835 -> 6 734 450,
 *       the size of water_caustics/water.obj), y = 1.45 + 0.06 * smooth value noise (3 octaves), no normals
 *       (the scene sets "smooth": Scene::generateVertexNormals, scene.cpp:61-65)
 *   gen_mesh bowl <nu> <nv> <cx> <cy> <cz> <rx> <ry> <rz> <dirx> <dirz> <seed> <out.obj>
 *       open ellipsoidal shell (2*nu*nv triangles): the cap of the ellipsoid around the horizontal direction
 *       (dirx, 0, dirz), radially displaced by +-4 % value noise — hull stand-ins for spaceship/aluminium.obj
 *       (500 x 300 -> 300 000) and spaceship/steel.obj (220 x 201 -> 88 440)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double hash3(int64_t ix, int64_t iy, int64_t iz, uint32_t seed) {
    uint64_t h = ((uint64_t)ix * 73856093ull) ^ ((uint64_t)iy * 19349663ull) ^ ((uint64_t)iz * 83492791ull) ^ (uint64_t)(seed * 2654435761u);
    h ^= h >> 15;
    h = (h * 0xd168aaadull) & 0xFFFFFFFFull;
    h ^= h >> 15;
    h = (h * 0xaf723597ull) & 0xFFFFFFFFull;
    h ^= h >> 15;
    return (double)(h & 0xFFFFFFull) / 16777216.0;
}

static double value_noise(double x, double y, double z, double freq, uint32_t seed) {
    const double qx = x * freq + 100.0, qy = y * freq + 100.0, qz = z * freq + 100.0;
    const int64_t ix = (int64_t)floor(qx), iy = (int64_t)floor(qy), iz = (int64_t)floor(qz);
    double fx = qx - (double)ix, fy = qy - (double)iy, fz = qz - (double)iz;
    fx = fx * fx * (3.0 - 2.0 * fx);
    fy = fy * fy * (3.0 - 2.0 * fy);
    fz = fz * fz * (3.0 - 2.0 * fz);
    double out = 0.0;
    for (int dx = 0; dx < 2; dx++)
        for (int dy = 0; dy < 2; dy++)
            for (int dz = 0; dz < 2; dz++) {
                const double w = (dx ? fx : 1.0 - fx) * (dy ? fy : 1.0 - fy) * (dz ? fz : 1.0 - fz);
                out += w * hash3(ix + dx, iy + dy, iz + dz, seed);
            }
    return out;
}

static int water(int cells, const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    static char buf[1 << 22];
    setvbuf(f, buf, _IOFBF, sizeof(buf));
    const int n = cells + 1;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            const double x = -1.0 + 2.0 * (double)i / (double)cells, z = -1.0 + 2.0 * (double)j / (double)cells;
            const double h = 0.5 * value_noise(x, 0.0, z, 3.0, 7u) + 0.3 * value_noise(x, 0.0, z, 7.0, 8u) + 0.2 * value_noise(x, 0.0, z, 17.0, 9u);
            fprintf(f, "v %.9g %.9g %.9g\n", x, 1.45 + 0.06 * (2.0 * h - 1.0), z);
        }
    for (int j = 0; j < cells; j++)
        for (int i = 0; i < cells; i++) {
            const long a = (long)j * n + i + 1, b = a + 1, c = a + n, d = c + 1;
            fprintf(f, "f %ld %ld %ld\nf %ld %ld %ld\n", a, c, b, b, c, d);
        }
    return fclose(f);
}

static int bowl(int nu, int nv, const double c[3], const double r[3], double dirx, double dirz, uint32_t seed, const char* path) {
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    static char buf[1 << 22];
    setvbuf(f, buf, _IOFBF, sizeof(buf));
    const double dl = sqrt(dirx * dirx + dirz * dirz), wx = dirx / dl, wz = dirz / dl;  /* cap axis (horizontal) */
    const double ux = -wz, uz = wx;                                                   /* horizontal tangent  */
    for (int j = 0; j <= nv; j++)
        for (int i = 0; i <= nu; i++) {
            const double a = 2.2 * (2.0 * (double)i / (double)nu - 1.0), b = 1.6 * (2.0 * (double)j / (double)nv - 1.0);
            /* gnomonic cap: direction = normalize(axis + a * tangent + b * up) */
            double px = wx + a * ux, py = b, pz = wz + a * uz;
            const double l = sqrt(px * px + py * py + pz * pz);
            px /= l; py /= l; pz /= l;
            const double s = 1.0 + 0.04 * (2.0 * value_noise(px, py, pz, 6.0, seed) - 1.0);
            fprintf(f, "v %.9g %.9g %.9g\n", c[0] + r[0] * s * px, c[1] + r[1] * s * py, c[2] + r[2] * s * pz);
        }
    for (int j = 0; j < nv; j++)
        for (int i = 0; i < nu; i++) {
            const long a = (long)j * (nu + 1) + i + 1, b = a + 1, cc = a + nu + 1, d = cc + 1;
            fprintf(f, "f %ld %ld %ld\nf %ld %ld %ld\n", a, b, cc, b, d, cc);
        }
    return fclose(f);
}

int main(int argc, char** argv) {
    if (argc == 4 && strcmp(argv[1], "water") == 0) return water(atoi(argv[2]), argv[3]);
    if (argc == 14 && strcmp(argv[1], "bowl") == 0) {
        const double c[3] = {atof(argv[4]), atof(argv[5]), atof(argv[6])}, r[3] = {atof(argv[7]), atof(argv[8]), atof(argv[9])};
        return bowl(atoi(argv[2]), atoi(argv[3]), c, r, atof(argv[10]), atof(argv[11]), (uint32_t)strtoul(argv[12], NULL, 0), argv[13]);
    }
    fprintf(stderr, "usage: gen_mesh water <cells> <out.obj> | gen_mesh bowl <nu> <nv> <cx> <cy> <cz> <rx> <ry> <rz> <dirx> <dirz> <seed> <out.obj>\n");
    return 2;
}
