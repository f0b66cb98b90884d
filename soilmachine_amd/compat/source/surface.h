// compat/source/surface.h -- drop-in for the reference's source/surface.h (surface.h:9-105).
// Same global tables (`soils`, `soilmap`, `layers`, `phong`) and the same SurfParam / SurfLayer members, so
// SoilMachine.cpp's GUI code and io.h keep compiling. The noise object of SurfLayer is gone: terrain
// initialisation (SurfLayer::get + FastNoiseLite) runs on the device (smx_initialize).
#pragma once
#include <map>
#include <string>
#include <vector>

using SurfType = size_t;

struct SurfParam {
  std::string name;
  float density;
  float porosity = 0.0f;
  glm::vec4 color = glm::vec4(0.5, 0.5, 0.5, 1.0);
  glm::vec4 phong = glm::vec4(0.5, 0.8, 0.2, 32);
  SurfType transports = 0;
  float solubility = 1.0f;
  float equrate = 1.0f;
  float friction = 1.0f;
  SurfType erodes = 0;
  float erosionrate = 0.0f;
  SurfType cascades = 0;
  float maxdiff = 1.0f;
  float settling = 0.0f;
  SurfType abrades = 0;
  float suspension = 0.0f;
  float abrasion = 0.0f;
};

std::vector<SurfParam> soils = {
    {"Air", 0.0f, 1.0f, glm::vec4(0.0, 0.2, 0.4, 1.0), glm::vec4(0.5, 0.8, 0.2, 32), 0, 0.0f, 0.0f, 0.0f, 0, 0.0f, 0, 0.0f, 0.0f, 0, 0.0f, 0.0f}};
std::map<std::string, int> soilmap = {{"Air", 0}};

struct SurfLayer {
  SurfType type;
  float min = 0.0f;
  float bias = 0.0f;
  float scale = 1.0f;
  float octaves = 1.0f;
  float lacunarity = 1.0f;
  float gain = 0.0f;
  float frequency = 1.0f;
  SurfLayer(const SurfType _type) { type = _type; }
};

std::vector<SurfLayer> layers;
std::vector<glm::vec4> phong;
