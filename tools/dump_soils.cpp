// Test helper: run the compat loadsoil() (soilmachine_amd/compat/source/io.h) and print the table as JSON.
#include <TinyEngine/TinyEngine>
#include <TinyEngine/image>
int SIZEX = 256, SIZEY = 256, SCALE = 80, SLICE = 160, NWIND = 250, NWATER = 250, SEED = 0;
#define POOLSIZE 10000000
#include "source/include/vertexpool.h"
#include "source/layermap.h"
#include "source/io.h"
int main(int argc, char** argv) {
  std::streambuf* old = std::cout.rdbuf(nullptr);
  loadsoil(argv[1]);
  std::cout.rdbuf(old);
  printf("{\"scale\": %d, \"sizex\": %d, \"sizey\": %d, \"nwater\": %d, \"nwind\": %d, \"soils\": [", SCALE, SIZEX, SIZEY, NWATER, NWIND);
  for (size_t i = 0; i < soils.size(); i++) {
    const SurfParam& p = soils[i];
    printf("%s{\"name\": \"%s\", \"transports\": %zu, \"erodes\": %zu, \"cascades\": %zu, \"abrades\": %zu, \"porosity\": %.9g, "
           "\"solubility\": %.9g, \"equrate\": %.9g, \"friction\": %.9g, \"erosionrate\": %.9g, \"maxdiff\": %.9g, \"settling\": %.9g, "
           "\"suspension\": %.9g, \"abrasion\": %.9g}", i ? ", " : "", p.name.c_str(), p.transports, p.erodes, p.cascades, p.abrades,
           p.porosity, p.solubility, p.equrate, p.friction, p.erosionrate, p.maxdiff, p.settling, p.suspension, p.abrasion);
  }
  printf("], \"layers\": [");
  for (size_t i = 0; i < layers.size(); i++) {
    const SurfLayer& l = layers[i];
    printf("%s{\"type\": %zu, \"min\": %.9g, \"bias\": %.9g, \"scale\": %.9g, \"octaves\": %.9g, \"lacunarity\": %.9g, \"gain\": %.9g, \"frequency\": %.9g}",
           i ? ", " : "", l.type, l.min, l.bias, l.scale, l.octaves, l.lacunarity, l.gain, l.frequency);
  }
  printf("]}\n");
  return 0;
}
