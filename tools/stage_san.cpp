// Sanitizer harness (CPU only): stage.cpp's scene staging incl. the uniform grid and the per-light visibility lists on POD scene files.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kajo_scene.h"
#include "stage.h"
int main(int argc, char** argv)
{
    for (int a = 1; a < argc; a++) {
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
        for (int lists = 0; lists < 2; lists++) {
            kajo::StagedScene out;
            kajo::stageScene(sc, out, 48, lists != 0);
            printf("%s: %d spheres %d planes, lists %d: ok\n", argv[a], n[0], n[1], lists);
        }
    }
    return 0;
}
