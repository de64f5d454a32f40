// Sanitizer harness (CPU only; tests/test_sanitize_cpu.py, tools/stage_sanitize.sh): the host-side index arithmetic of the product, compiled
// from the product's own sources with -fsanitize=address,undefined.
//   host_san stage  a.pod ...            kajo_amd/csrc/stage.cpp: object records, uniform grid, per-light visibility lists, coordinate range
//   host_san parse  aspect a.json ...    kajo_amd/host/scene/SceneLoader.cpp on scene files (and on every prefix of each: truncated input)
//   host_san order  in.bin out.bin       kajo_amd/csrc/launch_order.h: cost order + parted tail from a trip table; side-buffer slots checked
//   host_san tiles  W H tileW tileH owners   render_args.h kajoTileSlot over a whole frame: every slot in range, none taken twice
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kajo_scene.h"
#include "launch_order.h"
#include "render_args.h"
#include "scene/Scene.h"
#include "stage.h"

static int stage(int argc, char** argv)
{
    for (int a = 0; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) return 2;
        int32_t n[2];
        if (fread(n, 4, 2, f) != 2) return 3;
        KajoScene sc{};
        std::vector<KajoSphere> sp(n[0]);
        std::vector<KajoPlane> pl(n[1]);
        if (fread(sc.backgroundColor, 4, 4, f) != 4 || fread(&sc.camera, 4, 32, f) != 32) return 4;
        if (n[0] && fread(sp.data(), sizeof(KajoSphere), n[0], f) != (size_t)n[0]) return 5;
        if (n[1] && fread(pl.data(), sizeof(KajoPlane), n[1], f) != (size_t)n[1]) return 6;
        fclose(f);
        sc.nSpheres = n[0]; sc.nPlanes = n[1]; sc.spheres = sp.data(); sc.planes = pl.data();
        float lo, hi;
        kajo::coordinateRange(sc, &lo, &hi);
        for (int lists = 0; lists < 2; lists++) {
            kajo::StagedScene out;
            kajo::stageScene(sc, out, 48, lists != 0);
            printf("%s: %d spheres %d planes, lists %d, coordinates %g .. %g: ok\n", argv[a], n[0], n[1], lists, lo, hi);
        }
    }
    return 0;
}

static int parse(int argc, char** argv)
{
    if (argc < 2) return 2;
    const float aspect = (float)atof(argv[0]);
    for (int a = 1; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) return 3;
        std::string text;
        char buf[4096];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0)
            text.append(buf, got);
        fclose(f);
        scene::Scene s;
        if (!scene::Parser::loadFromString(s, text, aspect)) return 4;
        printf("%s: %zu spheres %zu planes\n", argv[a], s.spheres.size(), s.planes.size());
        // truncated input: the reader must fail or succeed, never read past the end (every prefix, in steps of 7 bytes)
        size_t accepted = 0;
        for (size_t cut = 0; cut < text.size(); cut += 7) {
            scene::Scene t;
            std::string part(text.data(), cut); // (a fresh, exactly sized buffer: an overread is an ASan error)
            accepted += scene::Parser::loadFromString(t, part, aspect) ? 1 : 0;
        }
        printf("%s: %zu of its prefixes parse\n", argv[a], accepted);
    }
    return 0;
}

static int order(int argc, char** argv)
{
    if (argc != 2) return 2;
    FILE* f = fopen(argv[0], "rb");
    if (!f) return 3;
    uint32_t head[5]; // nBlocks, wavesPerBlock, waveSlots, parts, threads
    if (fread(head, 4, 5, f) != 5) return 4;
    std::vector<uint32_t> trips((size_t)head[0] * head[1]);
    if (!trips.empty() && fread(trips.data(), 4, trips.size(), f) != trips.size()) return 5;
    fclose(f);
    std::vector<uint32_t> cost, plain, out;
    kajoBlockCosts(trips.data(), head[0], head[1], cost);
    kajoCostOrder(cost, plain);
    const unsigned nParted = kajoTailBlocks(head[0], head[2] / head[1]);
    const uint32_t parts = head[3], threads = head[4];
    if (parts >= 2 && nParted)
        kajoPartedOrder(plain, nParted, (int)parts, out);
    else
        out = plain;
    // side-buffer slots of the later parts' workgroups (render_args.h kajoSideSlot): in range, none taken twice
    if (parts >= 2 && nParted) {
        const uint32_t sideStride = nParted * threads, partedFirst = head[0] - nParted;
        std::vector<unsigned char> taken((size_t)(parts - 1) * sideStride, 0);
        for (size_t i = 0; i < out.size(); i++) {
            const uint32_t w = out[i];
            if (!(w & KAJO_ORDER_PARTED)) continue;
            if (i < partedFirst) return 6;
            const uint32_t part = (w >> KAJO_ORDER_PART_SHIFT) & 7u;
            if (part == 0) continue;
            for (uint32_t t = 0; t < threads; t += threads - 1 ? threads - 1 : 1) { // first and last thread of the workgroup
                const uint32_t s = kajoSideSlot((uint32_t)i, partedFirst, parts, part, sideStride, threads, t);
                if (s >= taken.size() || taken[s]) return 7;
                taken[s] = 1;
            }
        }
    }
    f = fopen(argv[1], "wb");
    if (!f) return 8;
    const uint32_t tail[2] = {nParted, (uint32_t)out.size()};
    fwrite(tail, 4, 2, f);
    if (!out.empty())
        fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    return 0;
}

static int tiles(int argc, char** argv)
{
    if (argc != 5) return 2;
    TileMap m{};
    m.W = atoi(argv[0]); m.H = atoi(argv[1]); m.tileW = atoi(argv[2]); m.tileH = atoi(argv[3]); m.tileCount = atoi(argv[4]);
    m.tilesX = (m.W + m.tileW - 1) / m.tileW;
    const int tilesY = (m.H + m.tileH - 1) / m.tileH, nTiles = m.tilesX * tilesY, perOwner = (nTiles + m.tileCount - 1) / m.tileCount;
    m.slotsPerOwner = perOwner * m.tileW * m.tileH;
    std::vector<unsigned char> taken((size_t)m.tileCount * m.slotsPerOwner, 0);
    for (int y = 0; y < m.H; y++)
        for (int x = 0; x < m.W; x++) {
            int owner;
            uint32_t slot;
            kajoTileSlot(m, x, y, &owner, &slot);
            if (owner < 0 || owner >= m.tileCount || slot >= (uint32_t)m.slotsPerOwner) return 3;
            unsigned char& t = taken[(size_t)owner * m.slotsPerOwner + slot];
            if (t) return 4;
            t = 1;
        }
    printf("%d x %d, tiles %d x %d, %d owners: %d slots per owner, ok\n", m.W, m.H, m.tileW, m.tileH, m.tileCount, m.slotsPerOwner);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    const std::string mode = argv[1];
    if (mode == "stage") return stage(argc - 2, argv + 2);
    if (mode == "parse") return parse(argc - 2, argv + 2);
    if (mode == "order") return order(argc - 2, argv + 2);
    if (mode == "tiles") return tiles(argc - 2, argv + 2);
    return 1;
}
