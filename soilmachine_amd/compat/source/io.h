// compat/source/io.h -- drop-in for the reference's source/io.h (io.h:7-252): the `.soil` parser and the two
// image exporters. The parser is restated WITH its quirks because they decide the soil
// indices the device tables use: one SurfParam value is reused for every SOIL block (unset keys are inherited),
// forward references allocate the index at first mention, names are val.substr(0, find('{') - 1).
// Pinned by tests/test_soilfile.py (Python twin) and tests/test_host_compat.py against the reference's loadsoil.
#pragma once
#include <fstream>
#include <functional>

void loadsoil(string file = "soil/default.soil") {
  ifstream in(file, ios::in);
  if (!in.is_open()) { cout << "Error: Failed to open soil profile " << file << endl; exit(0); }
  string line;
  int linenr = 0;
  auto syntaxerr = [&]() { cout << "Error: Incorrect Syntax in Line " << linenr << endl; exit(0); };
  auto hexcol = [&](string h) {
    if (h.size() < 6) syntaxerr();
    const string allowed = "0123456789ABCDEF";
    for (auto& c : h) if (allowed.find(c) == string::npos) syntaxerr();
    float R = 16 * allowed.find(h[0]) + allowed.find(h[1]);
    float G = 16 * allowed.find(h[2]) + allowed.find(h[3]);
    float B = 16 * allowed.find(h[4]) + allowed.find(h[5]);
    return vec4(R, G, B, 255.0) / 255.0f;
  };
  auto index_of = [&](const string& name, const SurfParam& proto) {     // first mention allocates (io.h:125-152)
    if (!soilmap.contains(name)) { soilmap[name] = soils.size(); soils.push_back(proto); }
    return (SurfType)soilmap[name];
  };

  SurfParam param;                      // ONE value for all SOIL blocks (io.h:35)
  bool open = false;
  string soillayer;
  while (getline(in, line)) {
    linenr++;
    size_t found = line.find('#');
    if (found != string::npos) line = line.substr(0, found);
    if (line == "") continue;
    if (line == "}") {
      if (!open) syntaxerr();
      if (soillayer == "SOIL") { cout << "Adding Soil Type " << param.name << endl; soils[soilmap[param.name]] = param; }
      open = false;
      continue;
    }
    found = line.find(' ');
    if (found == string::npos) syntaxerr();
    string tag = line.substr(0, found);
    string val = line.substr(found + 1);
    if (tag == "SOIL" || tag == "LAYER" || tag == "WORLD") {
      found = val.find('{');
      if (found == string::npos) syntaxerr();
      if (tag == "SOIL") { param.name = val.substr(0, found - 1); index_of(param.name, param); }
      if (tag == "LAYER") {
        param.name = val.substr(0, found - 1);
        if (!soilmap.contains(param.name)) { cout << "Can't find SOIL " << param.name << endl; syntaxerr(); }
        cout << "Adding Layer Type " << param.name << endl;
        layers.emplace_back(soilmap[param.name]);
      }
      soillayer = tag;
      open = true;
      continue;
    }
    if (soillayer == "SOIL") {
      if (tag == "TRANSPORTS") param.transports = index_of(val, param);
      if (tag == "ERODES") param.erodes = index_of(val, param);
      if (tag == "CASCADES") param.cascades = index_of(val, param);
      if (tag == "ABRADES") param.abrades = index_of(val, param);
      if (tag == "DENSITY") param.density = stof(val);
      if (tag == "POROSITY") param.porosity = stof(val);
      if (tag == "COLOR") param.color = hexcol(val);
      if (tag == "SOLUBILITY") param.solubility = stof(val);
      if (tag == "EQUILIBRIUM") param.equrate = stof(val);
      if (tag == "FRICTION") param.friction = stof(val);
      if (tag == "EROSIONRATE") param.erosionrate = stof(val);
      if (tag == "MAXDIFF") param.maxdiff = stof(val);
      if (tag == "SETTLING") param.settling = stof(val);
      if (tag == "SUSPENSION") param.suspension = stof(val);
      if (tag == "ABRASION") param.abrasion = stof(val);
      if (tag == "Ka") param.phong.x = stof(val);
      if (tag == "Kd") param.phong.y = stof(val);
      if (tag == "Ks") param.phong.z = stof(val);
      if (tag == "Kk") param.phong.w = stof(val);
    }
    if (soillayer == "LAYER") {
      if (tag == "MIN") layers.back().min = stof(val);
      if (tag == "BIAS") layers.back().bias = stof(val);
      if (tag == "SCALE") layers.back().scale = stof(val);
      if (tag == "OCTAVES") layers.back().octaves = stof(val);
      if (tag == "LACUNARITY") layers.back().lacunarity = stof(val);
      if (tag == "GAIN") layers.back().gain = stof(val);
      if (tag == "FREQUENCY") layers.back().frequency = stof(val);
    }
    if (soillayer == "WORLD") {
      if (tag == "SIZEX") SIZEX = stoi(val);
      if (tag == "SIZEY") SIZEY = stoi(val);
      if (tag == "SCALE") SCALE = stoi(val);
      if (tag == "NWIND") NWIND = stoi(val);
      if (tag == "NWATER") NWATER = stoi(val);
    }
  }
  in.close();
  for (size_t i = 0; i < soils.size(); i++) phong.push_back(soils[i].phong);
}

void exportcolor(Layermap& map, Vertexpool<Vertex>& vertexpool, string filename = "color.png") {   // io.h:234-242
  cout << "Exporting Color Image" << endl;
  map.update(vertexpool);
  SDL_Surface* img = image::make([&](ivec2 i) {
    Vertex* v = vertexpool.get(map.section, i.x * SIZEY + i.y);
    return vec4(v->color[2], v->color[1], v->color[0], 1);
  }, ivec2(SIZEX, SIZEY));
  image::save(img, filename);
}

void exportheight(Layermap& map, Vertexpool<Vertex>& vertexpool, string filename = "height.png") {   // io.h:245-252
  cout << "Exporting Height Image" << endl;
  map.update(vertexpool);
  SDL_Surface* img = image::make([&](ivec2 i) {
    Vertex* v = vertexpool.get(map.section, i.x * SIZEY + i.y);
    return vec4(v->position[1] / SCALE / sqrt(2), v->position[1] / SCALE / sqrt(2), v->position[1] / SCALE / sqrt(2), 1);
  }, ivec2(SIZEX, SIZEY));
  image::save(img, filename);
}
