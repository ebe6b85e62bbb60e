// filter_mrc (MI355X edition) -- the hot-path subset of the reference's filter_mrc command line,
// running on libvisfd_hip.so through the visfd:: shim (include/visfd_hip.hpp).
//
// Supported flags (same spelling, units and defaults as bin/filter_mrc/settings.cpp; all lengths are
// in PHYSICAL units and are divided by the voxel width, filter_mrc.cpp:297-336):
//   -in F | -i F        input tomogram (MRC modes 0,1,2,6)
//   -out F | -o F       output tomogram (always written as mode 2, like mrc_simple.cpp:373)
//   -mask F             mask tomogram (voxels with 0 are ignored)
//   -w WIDTH            voxel width (otherwise cellA[0]/nx from the header, handlers.cpp:2429)
//   -gauss S | -gauss-aniso SX SY SZ              (settings.cpp:1220-1271, HandleGauss)
//   -dog A B                                        (settings.cpp:1309-1335, HandleDog)
//   -log S | -log-r R | -log-d D | -log-aniso SX SY SZ | -dog-delta D   (HandleLoGDoG)
//   -blob|-blob-s|-blob-r|-blob-d TYPE FILE MIN MAX GROWTH             (settings.cpp:1648-1764)
//   -minima-threshold T | -maxima-threshold T                          (settings.cpp:1915,1934)
//   -membrane {minima|maxima} THICKNESS | -surface-ridge ...           (settings.cpp:2734-2799)
//   -tv RATIO | -tv-angle-exponent N | -tv-truncate R | -tv-best F | -detection-threshold T
//   -save-progress BASE        writes BASE_tensor_{0..5}.rec           (handlers.cpp:1897-1922)
//   -truncate R | -truncate-threshold T | -normalize-filters no | -bin 1 | -np N (ignored)
// Anything else is rejected, as the reference rejects unknown arguments (settings.cpp:3340-3365).
//
// MRC input/output is written from the MRC2014 layout description (1024-byte header: nx,ny,nz,mode,
// start[3], m[3], cella[3], cellb[3], mapc/r/s, dmin,dmax,dmean, ispg, nsymbt, ..., "MAP ", machst);
// signed-byte rule as the reference applies it (mrc_header.cpp:49-75, mrc_simple.cpp:186-192).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <sys/stat.h>
#include <ctime>
#include <thread>
#include <chrono>
#include <vector>

#include "../../include/visfd_hip.hpp"

using namespace visfd;
using std::cerr;
using std::string;
using std::vector;

namespace {

struct Mrc {
  int32_t nx = 0, ny = 0, nz = 0, mode = 2;
  float cella[3] = {0, 0, 0};
  unsigned char raw_header[1024];
  float*** a = nullptr;  // [iz][iy][ix], contiguous
  bool loaded = false;

  ~Mrc() { Dealloc3D(a); }

  void alloc(int x, int y, int z) {
    Dealloc3D(a);
    nx = x; ny = y; nz = z;
    int size[3] = {nx, ny, nz};
    a = Alloc3D<float>(size);
  }
  float* data() { return &a[0][0][0]; }
  size_t nvox() const { return (size_t)nx * ny * nz; }
  void swap(Mrc& o) {
    std::swap(nx, o.nx); std::swap(ny, o.ny); std::swap(nz, o.nz); std::swap(mode, o.mode);
    std::swap(a, o.a); std::swap(loaded, o.loaded);
    for (int d = 0; d < 3; d++) std::swap(cella[d], o.cella[d]);
    unsigned char t[1024];
    std::memcpy(t, raw_header, 1024); std::memcpy(raw_header, o.raw_header, 1024); std::memcpy(o.raw_header, t, 1024);
  }

  void read(const string& path) {
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) throw VisfdErr("Error: Unable to open \"" + path + "\" for reading.\n");
    f.read(reinterpret_cast<char*>(raw_header), 1024);
    if (!f) throw VisfdErr("Error: \"" + path + "\" is too short to be an MRC file.\n");
    int32_t w[256];
    std::memcpy(w, raw_header, 1024);
    float fw[256];
    std::memcpy(fw, raw_header, 1024);
    const int x = w[0], y = w[1], z = w[2];
    mode = w[3];
    if (x <= 0 || y <= 0 || z <= 0) throw VisfdErr("Error: bad image size in \"" + path + "\"\n");
    for (int d = 0; d < 3; d++) cella[d] = fw[10 + d];
    bool signed_bytes = true;
    if (path.size() > 4 && path.substr(path.size() - 4) == ".rec") signed_bytes = false;
    if (mode == 0 && w[38] == 1146047817) signed_bytes = (w[39] & 1) != 0;  // IMOD stamp + flag bit 0
    const int32_t nsymbt = w[23];
    if (nsymbt > 0) f.seekg(nsymbt, std::ios::cur);
    alloc(x, y, z);
    const size_t n = nvox();
    float* out = data();
    if (mode == 2) {
      f.read(reinterpret_cast<char*>(out), (std::streamsize)(n * 4));
    } else if (mode == 0) {
      vector<unsigned char> buf(n);
      f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)n);
      for (size_t i = 0; i < n; i++) out[i] = signed_bytes ? (float)(int8_t)buf[i] : (float)buf[i];
    } else if (mode == 1 || mode == 6) {
      vector<uint16_t> buf(n);
      f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(n * 2));
      for (size_t i = 0; i < n; i++) out[i] = (mode == 1) ? (float)(int16_t)buf[i] : (float)buf[i];
    } else {
      throw VisfdErr("Error: unsupported MRC mode in \"" + path + "\" (supported: 0, 1, 2, 6)\n");
    }
    if (!f) throw VisfdErr("Error: \"" + path + "\" ended before all voxels were read.\n");
    loaded = true;
  }

  // header of `like` (cell size, origin, labels) with mode 2 and fresh statistics
  void write(const string& path, const Mrc& like) {
    unsigned char h[1024];
    std::memcpy(h, like.raw_header, 1024);
    int32_t w[256];
    std::memcpy(w, h, 1024);
    float fw[256];
    std::memcpy(fw, h, 1024);
    if (nx != w[0] || ny != w[1] || nz != w[2]) {   // resized by binning: grid and cell follow the new size
      w[7] = nx; w[8] = ny; w[9] = nz;
      std::memcpy(&w[10], like.cella, 12);
    }
    w[0] = nx; w[1] = ny; w[2] = nz; w[3] = 2;
    w[23] = 0;  // no extended header
    const size_t n = nvox();
    const float* p = &a[0][0][0];
    double sum = 0;
    float lo = p[0], hi = p[0];
    for (size_t i = 0; i < n; i++) { sum += p[i]; lo = std::min(lo, p[i]); hi = std::max(hi, p[i]); }
    std::memcpy(h, w, 96);
    fw[19] = lo; fw[20] = hi; fw[21] = (float)(sum / (double)n);
    std::memcpy(h + 76, &fw[19], 12);
    std::ofstream f(path.c_str(), std::ios::binary);
    if (!f) throw VisfdErr("Error: Unable to open \"" + path + "\" for writing.\n");
    f.write(reinterpret_cast<const char*>(h), 1024);
    f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * 4));
  }
};

struct Settings {
  string in, out, mask, save_base;
  float voxel_width = -1;
  int bin = 0;                 // settings.cpp:48-49: 0 = not specified (automatic), else the factor
  bool bin_explicit = false;
  float masked_voxel_brightness = 0.0f;   // settings.cpp:41-42: voxels with mask == 0 get this value in the output
  enum { NONE, GAUSS, DOG, LOG, BLOB, BLOB_NONMAX, SURFACE_RIDGE, LOCAL_FLUCTUATIONS } type = NONE;
  float width_a[3] = {0, 0, 0}, width_b[3] = {0, 0, 0}, log_width[3] = {0, 0, 0};
  float template_background_radius[3] = {-1, -1, -1};        // settings.cpp:222-225 (-fluct)
  float template_background_exponent = 2.0f;
  float truncate_ratio = -1.0f, truncate_threshold = 0.03f;   // settings.cpp:81,88
  float delta = 0.02f;                                        // settings.cpp:95
  bool normalize = true;
  // blobs
  vector<float> blob_diameters;
  float blob_aspect_ratio[3] = {1.0f, 1.0f, 1.0f};            // settings.cpp:134-136, -blob-aspect-ratio
  string blob_min_file, blob_max_file;
  float score_lower = -std::numeric_limits<float>::infinity();
  float score_upper = std::numeric_limits<float>::infinity();
  // blob list post-processing (-discard-blobs)
  vector<string> in_crds_files;
  string out_crds_file;
  float nonmax_min_radial_separation_ratio = 0.0f;            // settings.cpp:137
  float nonmax_max_overlap_large = std::numeric_limits<float>::infinity();
  float nonmax_max_overlap_small = std::numeric_limits<float>::infinity();
  // clustering of the detected surface (-connect ...), settings.cpp:163-178
  bool cluster_connected_voxels = false;
  string must_link_filename;                       // -must-link FILE (settings.cpp:3183-3195)
  vector<float> must_link_crds;                    // flat x,y,z of every location, group after group
  vector<int64_t> must_link_group_sizes;
  vector<int> must_link_directions;                // 0 same, 1 opposite, 2 automatic (one per location)
  bool must_link_in_voxels = false;
  float connect_threshold_saliency = std::numeric_limits<float>::infinity();
  float connect_threshold_vector_saliency = (float)std::cos(M_PI * 15 / 180.0);
  float connect_threshold_vector_neighbor = (float)std::cos(M_PI * 15 / 180.0);
  float connect_threshold_tensor_saliency = (float)std::cos(M_PI * 15 / 180.0);
  float connect_threshold_tensor_neighbor = (float)std::cos(M_PI * 15 / 180.0);
  string out_normals_file;                                    // -normals-file (settings.cpp:2965-2979)
  int select_cluster = 0;                                     // settings.cpp:168
  float max_distance_to_feature = 1.3f;                       // settings.cpp:147 (voxels; negative: physical units)
  float surface_normal_curve_ds = 0.2f;                       // settings.cpp:148
  bool surface_find_ridge = true;                             // settings.cpp:149
  bool undefined_voxels_are_max = true;                       // settings.cpp:43-44
  float undefined_voxel_brightness = -1.0f;
  string load_base;
  // membranes
  bool ridges_are_maxima = false;
  float hessian_thr = 0.05f;                                  // settings.cpp:150-151
  bool hessian_thr_is_fraction = true;
  float tv_sigma = 0.0f;
  int tv_exponent = 4;                                        // settings.cpp:154
  float tv_truncate = std::sqrt(2.0);                         // settings.cpp:155
  // Z-slab run across the GPUs of a node (no reference counterpart: the reference is single-process).  One filter_mrc per GPU:
  //   VISFD_HIP_DEVICE=r filter_mrc ... -slab r WORLD IDFILE -out out_r.rec
  int slab_rank = -1, slab_world = 0;
  string slab_id_file;
};

bool read_must_link_file(const string& path, Settings& s);

float num(const vector<string>& v, size_t i, const string& flag) {
  if (i >= v.size() || v[i].empty()) throw VisfdErr("Error: The " + flag + " argument must be followed by a number.\n");
  try { return std::stof(v[i]); } catch (...) { throw VisfdErr("Error: The " + flag + " argument must be followed by a number.\n"); }
}

Settings parse(int argc, char** argv) {
  Settings s;
  vector<string> v(argv + 1, argv + argc);
  for (size_t i = 0; i < v.size();) {
    const string& f = v[i];
    auto need = [&](size_t k) { if (i + k >= v.size()) throw VisfdErr("Error: The " + f + " argument needs " + std::to_string(k) + " parameter(s).\n"); };
    if (f == "-in" || f == "-i") { need(1); s.in = v[i + 1]; i += 2; }
    else if (f == "-out" || f == "-o") { need(1); s.out = v[i + 1]; i += 2; }
    else if (f == "-mask") { need(1); s.mask = v[i + 1]; i += 2; }
    else if (f == "-w") { need(1); s.voxel_width = num(v, i + 1, f); i += 2; }
    else if (f == "-mask-out") { need(1); s.masked_voxel_brightness = num(v, i + 1, f); i += 2; }   // settings.cpp:662-674
    else if (f == "-np") { need(1); i += 2; }  // host threads: not used by the GPU path
    else if (f == "-bin") {
      need(1);
      const float b = num(v, i + 1, f);
      if (b < 1.0f || b != std::floor(b)) throw VisfdErr("Error: The " + f + " argument must be followed by a positive integer.\n");
      s.bin = (int)b;              // settings.cpp:703-716
      s.bin_explicit = true;
      i += 2;
    }
    else if (f == "-gauss") { need(1); s.width_a[0] = s.width_a[1] = s.width_a[2] = num(v, i + 1, f); s.type = Settings::GAUSS; i += 2; }
    else if (f == "-fluct" || f == "-fluctuation" || f == "-fluctuations") {     // settings.cpp:2170-2186
      need(1);
      s.template_background_radius[0] = s.template_background_radius[1] = s.template_background_radius[2] = num(v, i + 1, f);
      s.type = Settings::LOCAL_FLUCTUATIONS; s.masked_voxel_brightness = 0.0f; i += 2;
    }
    else if (f == "-fluct-aniso" || f == "-fluctuation-aniso" || f == "-fluctuations-aniso") {   // settings.cpp:2138-2156
      need(3);
      for (int d = 0; d < 3; d++) s.template_background_radius[d] = num(v, i + 1 + d, f);
      s.type = Settings::LOCAL_FLUCTUATIONS; s.masked_voxel_brightness = 0.0f; i += 4;
    }
    else if (f == "-gauss-aniso") { need(3); for (int d = 0; d < 3; d++) s.width_a[d] = num(v, i + 1 + d, f); s.type = Settings::GAUSS; i += 4; }
    else if (f == "-dog") {
      need(2);
      s.width_a[0] = s.width_a[1] = s.width_a[2] = num(v, i + 1, f);
      s.width_b[0] = s.width_b[1] = s.width_b[2] = num(v, i + 2, f);
      s.type = Settings::DOG; i += 3;
    }
    else if (f == "-dog-aniso") {                                                         // settings.cpp:1275-1305
      if (i + 6 >= v.size()) throw VisfdErr("Error: The " + f + " argument must be followed by 6 positive numbers.\n");
      for (int k = 1; k <= 6; k++)
        if (v[i + k].empty() || v[i + k][0] == '-') throw VisfdErr("Error: The " + f + " argument must be followed by 6 positive numbers.\n");
      for (int d = 0; d < 3; d++) { s.width_a[d] = num(v, i + 1 + d, f); s.width_b[d] = num(v, i + 4 + d, f); }
      s.type = Settings::DOG; i += 7;
    }
    else if (f == "-blob-aspect-ratio") {                                                 // settings.cpp:1628-1644
      need(3);
      for (int d = 0; d < 3; d++) s.blob_aspect_ratio[d] = num(v, i + 1 + d, f);
      i += 4;
    }
    else if (f == "-log" || f == "-log-r" || f == "-log-d") {
      need(1);
      float m = 1.0f;
      if (f == "-log-r") m = (float)(1.0 / std::sqrt(3.0));
      if (f == "-log-d") m = (float)(1.0 / (2.0 * std::sqrt(3.0)));
      s.log_width[0] = s.log_width[1] = s.log_width[2] = num(v, i + 1, f) * m;
      s.type = Settings::LOG; i += 2;
    }
    else if (f == "-log-aniso") { need(3); for (int d = 0; d < 3; d++) s.log_width[d] = num(v, i + 1 + d, f); s.type = Settings::LOG; i += 4; }
    else if (f == "-dog-delta") { need(1); s.delta = num(v, i + 1, f); i += 2; }
    else if (f == "-truncate") { need(1); s.truncate_ratio = num(v, i + 1, f); s.truncate_threshold = -1.0f; i += 2; }
    else if (f == "-truncate-threshold") { need(1); s.truncate_threshold = num(v, i + 1, f); s.truncate_ratio = -1.0f; i += 2; }
    else if (f == "-normalize-filters") {
      need(1);
      if (v[i + 1] == "no") s.normalize = false;
      else throw VisfdErr("Error: -normalize-filters accepts \"no\" only (as in the reference, settings.cpp:492-496).\n");
      i += 2;
    }
    else if (f == "-blob" || f == "-blob-sigma" || f == "-blob-s" || f == "-blobs" || f == "-blob-radii" ||
             f == "-blob-r" || f == "-blobr" || f == "-blob-diameters" || f == "-blob-d") {
      need(5);
      const string kind = v[i + 1], base = v[i + 2];
      if (kind == "minima" || kind == "min") { s.blob_min_file = base; s.blob_max_file = ""; s.score_upper = 0.0f; }
      else if (kind == "maxima" || kind == "max") { s.blob_max_file = base; s.blob_min_file = ""; s.score_lower = 0.0f; }
      else if (kind == "all") {
        s.blob_min_file = base + ".minima.txt"; s.blob_max_file = base + ".maxima.txt";
        if (s.score_lower == 0.0f) s.score_lower = -std::numeric_limits<float>::infinity();
        if (s.score_upper == 0.0f) s.score_upper = std::numeric_limits<float>::infinity();
      } else throw VisfdErr("Error: The 1st parameter to \"" + f + "\" must be \"minima\", \"maxima\" or \"all\".\n");
      const float wmin = num(v, i + 3, f), wmax = num(v, i + 4, f);
      float growth = num(v, i + 5, f);
      if (wmin <= 0 || wmax <= 0 || wmin >= wmax || growth <= 1.0f)
        throw VisfdErr("Error: " + f + " needs 0 < min < max and a growth ratio > 1.\n");
      const int N = 1 + (int)std::ceil(std::log(wmax / wmin) / std::log(growth));   // settings.cpp:1719
      growth = (float)std::pow(wmax / wmin, 1.0 / N);
      float mult = 1.0f;
      if (f == "-blob-sigma" || f == "-blob-s") mult = (float)(2.0 * std::sqrt(3.0));
      if (f == "-blob-radii" || f == "-blob-r" || f == "-blobr") mult = 2.0f;
      else if (f == "-blob-diameters" || f == "-blob-d") mult = 1.0f;
      s.blob_diameters.resize((size_t)N);
      s.blob_diameters[0] = wmin * mult;
      for (int n = 1; n < N; n++) s.blob_diameters[(size_t)n] = s.blob_diameters[(size_t)n - 1] * growth;
      s.type = Settings::BLOB; i += 6;
    }
    else if (f == "-discard-blobs" || f == "-blob-nonmax" || f == "-blobs-nonmax") {   // settings.cpp:1769-1787
      need(2);
      if (v[i + 1].empty() || v[i + 1][0] == '-' || v[i + 2].empty() || v[i + 2][0] == '-' || v[i + 1] == v[i + 2])
        throw VisfdErr("Error: The " + f + " argument must be followed by two different file names\n");
      s.in_crds_files.push_back(v[i + 1]);
      s.out_crds_file = v[i + 2];
      s.type = Settings::BLOB_NONMAX;
      i += 3;
    }
    else if (f == "-radial-separation" || f == "-blob-separation" || f == "-blob-r-separation" ||
             f == "-blobr-separation" || f == "-spheres-nonmax-separation-radius") {   // settings.cpp:1603-1624
      need(1); s.nonmax_min_radial_separation_ratio = num(v, i + 1, f); i += 2;
    }
    else if (f == "-max-volume-overlap") { need(1); s.nonmax_max_overlap_large = num(v, i + 1, f); i += 2; }        // settings.cpp:1540
    else if (f == "-max-volume-overlap-small") { need(1); s.nonmax_max_overlap_small = num(v, i + 1, f); i += 2; }  // settings.cpp:1561
    else if (f == "-minima-threshold") { need(1); s.score_upper = num(v, i + 1, f); i += 2; }
    else if (f == "-maxima-threshold") { need(1); s.score_lower = num(v, i + 1, f); i += 2; }
    else if (f == "-membrane" || f == "-surface-ridge") {
      need(2);
      if (v[i + 1] == "min" || v[i + 1] == "minima") s.ridges_are_maxima = false;
      else if (v[i + 1] == "max" || v[i + 1] == "maxima") s.ridges_are_maxima = true;
      else throw VisfdErr("Error: The " + f + " argument must be followed by \"minima\" or \"maxima\" and a width.\n");
      const float sigma = (float)(num(v, i + 2, f) / std::sqrt(3.0));   // settings.cpp:2774
      s.width_a[0] = s.width_a[1] = s.width_a[2] = sigma;
      s.type = Settings::SURFACE_RIDGE; i += 3;
    }
    else if (f == "-detection-background" || f == "-membrane-background" || f == "-curve-background") {   // settings.cpp:2802-2825
      // the peak-height factor of the score loops: width (sigma, physical units) of the Gaussian whose output is the
      // background; both scores are multiplied by (image - background), handlers.cpp:1577-1605,1698-1702,1883-1887
      need(1);
      s.width_b[0] = s.width_b[1] = s.width_b[2] = num(v, i + 1, f);
      s.type = Settings::SURFACE_RIDGE; i += 2;
    }
    else if (f == "-tv") { need(1); s.tv_sigma = num(v, i + 1, f); i += 2; }
    else if (f == "-tv-angle-exponent") { need(1); s.tv_exponent = (int)num(v, i + 1, f); i += 2; }
    else if (f == "-tv-truncate-ratio") { need(1); s.tv_truncate = num(v, i + 1, f); i += 2; }   // settings.cpp:2931-2946
    else if (f == "-tv-best" || f == "-best") {
      need(1); s.hessian_thr = num(v, i + 1, f); s.hessian_thr_is_fraction = true;
      if (!(s.hessian_thr >= 0.0f && s.hessian_thr <= 1.0f)) throw VisfdErr("Error: -tv-best needs a number between 0 and 1.\n");
      i += 2;
    }
    else if (f == "-detection-threshold") { need(1); s.hessian_thr = num(v, i + 1, f); s.hessian_thr_is_fraction = false; i += 2; }
    else if (f == "-slab") {
      need(3);
      s.slab_rank = (int)num(v, i + 1, f); s.slab_world = (int)num(v, i + 2, f); s.slab_id_file = v[i + 3];
      if (s.slab_world < 1 || s.slab_rank < 0 || s.slab_rank >= s.slab_world)
        throw VisfdErr("Error: -slab RANK WORLD IDFILE needs 0 <= RANK < WORLD.\n");
      i += 4;
    }
    else if (f == "-save-progress") { need(1); s.save_base = v[i + 1]; i += 2; }
    else if (f == "-load-progress") { need(1); s.load_base = v[i + 1]; i += 2; }
    // (-connect-dark differs from -connect only in clusters_begin_at_maxima, settings.cpp:3057-3060, a flag that nothing on
    //  the membrane path reads: handlers.cpp:1341 is its only use, in the watershed handler)
    else if (f == "-connect" || f == "-connect-bright" || f == "-connect-saliency" || f == "-connect-dark") {   // settings.cpp:3036-3072
      need(1); s.cluster_connected_voxels = true; s.connect_threshold_saliency = num(v, i + 1, f); i += 2;
    }
    else if (f == "-connect-angle") {                                                    // settings.cpp:3075-3094
      need(1); s.cluster_connected_voxels = true;
      const double theta = num(v, i + 1, f);
      const float c = (float)std::cos(theta * M_PI / 180.0);
      s.connect_threshold_vector_saliency = s.connect_threshold_vector_neighbor = c;
      s.connect_threshold_tensor_saliency = s.connect_threshold_tensor_neighbor = c;
      i += 2;
    }
    else if (f == "-must-link") { need(1); s.must_link_filename = v[i + 1]; i += 2; }
    else if (f == "-connect-vector-saliency") { need(1); s.cluster_connected_voxels = true; s.connect_threshold_vector_saliency = num(v, i + 1, f); i += 2; }
    else if (f == "-connect-vector-neighbor") { need(1); s.cluster_connected_voxels = true; s.connect_threshold_vector_neighbor = num(v, i + 1, f); i += 2; }
    else if (f == "-connect-tensor-saliency") { need(1); s.cluster_connected_voxels = true; s.connect_threshold_tensor_saliency = num(v, i + 1, f); i += 2; }
    else if (f == "-connect-tensor-neighbor") { need(1); s.cluster_connected_voxels = true; s.connect_threshold_tensor_neighbor = num(v, i + 1, f); i += 2; }
    else if (f == "-undefined-out") {                                                    // settings.cpp:2683-2700
      need(1);
      if (v[i + 1] == "max") s.undefined_voxels_are_max = true;
      else { s.undefined_voxels_are_max = false; s.undefined_voxel_brightness = num(v, i + 1, f); }
      i += 2;
    }
    else if (f == "-select-cluster") {                                                     // settings.cpp:3163-3182
      need(1); s.select_cluster = (int)num(v, i + 1, f); s.cluster_connected_voxels = true;
      if (s.select_cluster < 0) throw VisfdErr("Error: The " + f + " argument must be followed by a positive integer.\n");
      i += 2;
    }
    else if (f == "-normals-file" || f == "-surface-normals-file") { need(1); s.out_normals_file = v[i + 1]; i += 2; }
    else if (f == "-max-voxels-to-feature" || f == "-max-voxels-to-surface" || f == "-max-voxels-to-membrane") {   // settings.cpp:2983-3005
      need(1);
      const string a = v[i + 1];
      s.max_distance_to_feature = (a == "inf" || a == "infinity" || a == "disable") ? 0.0f : num(v, i + 1, f);
      i += 2;
    }
    else if (f == "-max-distance-to-feature" || f == "-max-distance-to-surface" || f == "-max-distance-to-membrane") {   // :3010-3032
      need(1);
      const string a = v[i + 1];
      s.max_distance_to_feature = (a == "inf" || a == "infinity" || a == "disable") ? 0.0f : -num(v, i + 1, f);
      i += 2;
    }
    else throw VisfdErr("Error: Unrecognized (or unsupported on the GPU hot path) argument: \"" + f + "\"\n");
  }
  if (s.in.empty()) throw VisfdErr("Error: You must specify an input file (-in).\n");
  if (!s.must_link_filename.empty()) s.must_link_in_voxels = read_must_link_file(s.must_link_filename, s);
  if (s.type == Settings::SURFACE_RIDGE) s.tv_sigma *= s.width_a[0];   // settings.cpp:3535-3540
  if (s.cluster_connected_voxels && s.type != Settings::SURFACE_RIDGE)
    throw VisfdErr("Error: this build clusters voxels (-connect) only after \"-membrane ... -tv ...\".\n");
  if (s.cluster_connected_voxels && s.connect_threshold_saliency == std::numeric_limits<float>::infinity())
    throw VisfdErr("Error: clustering needs a saliency threshold (-connect THRESHOLD).\n");
  if (!s.out_normals_file.empty() && !s.cluster_connected_voxels)
    throw VisfdErr("Error: this build writes surface normals (-normals-file) for a clustered surface only (-connect).\n");
  if ((s.cluster_connected_voxels || !s.load_base.empty()) && !(s.tv_sigma > 0))
    throw VisfdErr("Error: -connect and -load-progress need tensor voting (-tv).\n");
  return s;
}

// Blob list file (bin/filter_mrc/file_io.hpp:413-493): 3-5 numbers per line (x y z [diameter [score]]), '#'
// starts a comment; coordinates written IMOD-style in parentheses mean "units of voxels".  Returns that flag.
// The -must-link file (bin/filter_mrc/file_io.hpp:82-214, :667-747): groups of locations separated by blank lines; a
// line holds x y z and optionally a fourth number (> 0: the two surfaces face the same way, < 0: opposite, else
// automatic); text after '#' is ignored.  IMOD's notation -- "Pixel (x, y, z) = value" or any line whose numbers sit in
// parentheses -- means 1-based voxel indices: floor(x) - 1.  Returns whether the coordinates are voxels already.
bool read_must_link_file(const string& path, Settings& s) {
  std::ifstream f(path.c_str());
  if (!f) throw VisfdErr("Error: unable to open \"" + path + "\" for reading.\n");
  bool imod_any = false;
  vector<std::array<float, 3> > group;
  vector<int> group_dirs;
  auto close_group = [&]() {
    if (group.empty()) return;
    if (group.size() < 2 || group[0] == group[1])
      throw VisfdErr("Error: Format error in file \"" + path + "\".\n"
                     "       Each group must contain at least 2 voxels.  (Voxels appear on different\n"
                     "       lines, so blank-line delimters must not separate SINGLE non-blank lines)\n"
                     "       Furthermore, the voxels in each set must be unique.\n");
    s.must_link_group_sizes.push_back((int64_t)group.size());
    for (size_t k = 0; k < group.size(); k++) {
      for (int d = 0; d < 3; d++) s.must_link_crds.push_back(group[k][d]);
      s.must_link_directions.push_back(group_dirs[k]);
    }
    group.clear();
    group_dirs.clear();
  };
  string line;
  while (std::getline(f, line)) {
    const size_t hash = line.find('#');
    if (hash != string::npos) line = line.substr(0, hash);
    bool parens = false, imod = false;
    for (size_t k = 0; k < line.size(); k++) {
      if (line[k] == '(' || line[k] == ')') { parens = true; line[k] = ' '; }
      else if (line[k] == ',') line[k] = ' ';
    }
    std::istringstream ws(line);
    vector<string> words;
    string w;
    while (ws >> w) words.push_back(w);
    if (!words.empty() && words[0] == "Pixel") { imod = parens = true; words.erase(words.begin()); }
    vector<float> xyz;
    for (size_t d = 0; d < words.size(); d++) {
      if (d >= 3 && imod) break;                       // "= value" of IMOD's line
      std::istringstream num(words[d]);
      float x;
      if (!(num >> x)) throw VisfdErr("Error: File read error (invalid entry?) on line:\n      " + line + "\n");
      if (parens && xyz.size() < 3) x = std::floor(x) - 1.0f;   // IMOD counts voxels from 1
      xyz.push_back(x);
    }
    imod_any = imod_any || parens;
    if (xyz.empty()) { close_group(); continue; }
    if (xyz.size() != 3 && xyz.size() != 4)
      throw VisfdErr("Error: Each line of file \"" + path + "\"\n       should contain either 3 numbers, 4 numbers, or 0 numbers.\n");
    std::array<float, 3> c = {{xyz[0], xyz[1], xyz[2]}};
    group.push_back(c);
    group_dirs.push_back(xyz.size() == 4 ? (xyz[3] > 0 ? 0 : (xyz[3] < 0 ? 1 : 2)) : 2);
  }
  close_group();
  if (s.must_link_group_sizes.empty())
    throw VisfdErr("Error: Format error in file \"" + path + "\".\n       File contains no voxel coordinates.\n");
  return imod_any;
}

bool read_blob_file(const string& path, vector<std::array<float, 3> >& crds, vector<float>& diameters,
                    vector<float>& scores) {
  std::ifstream f(path.c_str());
  if (!f) throw VisfdErr("Error: unable to open \"" + path + "\" for reading.\n");
  bool parens = false;
  string line;
  size_t i_line = 0;
  while (std::getline(f, line)) {
    const size_t hash = line.find('#');
    if (hash != string::npos) line.erase(hash);
    for (size_t k = 0; k < line.size(); k++) {
      if (line[k] == '(' || line[k] == ')') { parens = true; line[k] = ' '; }
      else if (line[k] == ',') line[k] = ' ';
    }
    std::istringstream in(line);
    vector<float> nums;
    string tok;
    while (in >> tok) {
      try { nums.push_back(std::stof(tok)); } catch (...) { /* words such as "Pixel" or "=" are skipped */ }
    }
    if (nums.empty()) continue;
    if (nums.size() < 3 || nums.size() > 5) {
      std::ostringstream msg;
      msg << "Error: Error on line " << i_line + 1 << " of file \"" << path << "\"\n"
          << "       Each line should contain either 3-5 numbers, or 0 numbers (blank).\n";
      throw VisfdErr(msg.str());
    }
    std::array<float, 3> c = {{nums[0], nums[1], nums[2]}};
    crds.push_back(c);
    float d = nums.size() > 3 ? nums[3] : -1.0f;
    if (d < 0) d = -1.0f;
    diameters.push_back(d);
    scores.push_back(nums.size() > 4 ? nums[4] : 1.0f);   // default score = sphere_decals_foreground (settings.cpp: 1)
    i_line++;
  }
  return parens;
}

// HandleBlobsNonmaxSuppression, bin/filter_mrc/handlers.cpp:421-617 (without the supervised-learning tail)
void handle_blob_nonmax(const Settings& s, const float vw[3], float const* const* const* mask, const int size[3]) {
  const float w = vw[0];
  const float inf = std::numeric_limits<float>::infinity();
  vector<std::array<float, 3> > crds;
  vector<float> diameters, scores;
  for (size_t I = 0; I < s.in_crds_files.size(); I++) {
    vector<std::array<float, 3> > c;
    vector<float> d, sc;
    const bool in_voxels = read_blob_file(s.in_crds_files[I], c, d, sc);
    if (!in_voxels && w > 0.0f)
      for (size_t i = 0; i < c.size(); i++) {
        for (int k = 0; k < 3; k++) c[i][k] = (float)std::floor((c[i][k] / w) + 0.5);   // handlers.cpp:458
        if (d[i] != -1.0f) d[i] /= w;
      }
    crds.insert(crds.end(), c.begin(), c.end());
    diameters.insert(diameters.end(), d.begin(), d.end());
    scores.insert(scores.end(), sc.begin(), sc.end());
  }
  cerr << " --- discarding blobs in files ---\n\n";
  if (s.score_lower != -inf || s.score_upper != inf) {   // handlers.cpp:503-545
    vector<std::array<float, 3> > c;
    vector<float> d, sc;
    for (size_t i = 0; i < crds.size(); i++)
      if (scores[i] >= s.score_lower && scores[i] <= s.score_upper) {
        c.push_back(crds[i]); d.push_back(diameters[i]); sc.push_back(scores[i]);
      }
    crds.swap(c); diameters.swap(d); scores.swap(sc);
  }
  if (!crds.empty() && mask) {
    cerr << "  discarding blobs outside the mask" << std::endl;
    DiscardMaskedBlobs(crds, diameters, scores, mask, size);
  }
  if (s.nonmax_min_radial_separation_ratio > 0 || s.nonmax_max_overlap_large != inf || s.nonmax_max_overlap_small != inf) {
    if (w <= 0.0f)
      throw VisfdErr("Error: Checking for overlapping blobs requires that you either specify the\n"
                     "       voxel width (using the \"-w\" argument).\n");
    cerr << "  discarding overlapping blobs" << std::endl;
    DiscardOverlappingBlobs(crds, diameters, scores, s.nonmax_min_radial_separation_ratio, s.nonmax_max_overlap_large,
                            s.nonmax_max_overlap_small, SORT_DECREASING_MAGNITUDE, &cerr);
  }
  cerr << " " << crds.size() << " blobs remaining" << std::endl;
  if (!s.out_crds_file.empty()) {
    const double wp = w > 0.0f ? (double)w : 1.0;
    std::ofstream out(s.out_crds_file.c_str());
    if (!out) throw VisfdErr("Error: unable to open \"" + s.out_crds_file + "\" for writing.\n");
    for (size_t i = 0; i < crds.size(); i++)
      out << crds[i][0] * wp << " " << crds[i][1] * wp << " " << crds[i][2] * wp << " " << diameters[i] * wp << " "
          << scores[i] << std::endl;
  }
}

// HandleBinning, bin/filter_mrc/handlers.cpp:2361-2425: the image (and the mask) shrink by `bin` per axis
// (BinArray3D averages; trailing voxels are dropped) and the voxel width grows by the same factor.
void bin_image(Mrc& img, int bin, double voxel_width_binned) {
  int ssz[3] = {img.nx, img.ny, img.nz};
  int dsz[3] = {img.nx / bin, img.ny / bin, img.nz / bin};
  if (dsz[0] < 1 || dsz[1] < 1 || dsz[2] < 1) throw VisfdErr("Error: the image is too small for this bin size.\n");
  Mrc tmp;
  tmp.alloc(dsz[0], dsz[1], dsz[2]);
  BinArray3D(ssz, dsz, img.a, tmp.a);
  std::memcpy(tmp.raw_header, img.raw_header, 1024);
  tmp.mode = img.mode;
  tmp.loaded = true;
  for (int d = 0; d < 3; d++) tmp.cella[d] = (float)(voxel_width_binned * dsz[d]);
  img.swap(tmp);
}

// the tail of HandleTV (handlers.cpp:2315-2355): an image that was binned WITHOUT the user asking for it
// goes back to the original size (nearest-lower sampling)
void unbin_image(Mrc& img, const int size_orig[3], const float cella_orig[3]) {
  int ssz[3] = {img.nx, img.ny, img.nz};
  Mrc big;
  big.alloc(size_orig[0], size_orig[1], size_orig[2]);
  UnbinArray3D(ssz, size_orig, img.a, big.a);
  std::memcpy(big.raw_header, img.raw_header, 1024);
  big.mode = img.mode;
  big.loaded = true;
  for (int d = 0; d < 3; d++) big.cella[d] = cella_orig[d];
  img.swap(big);
}

float ratio_of(const Settings& s) {
  return s.truncate_ratio > 0 ? s.truncate_ratio : visfd_hip_ratio_from_threshold(s.truncate_threshold);
}

}  // namespace

// -membrane ... -tv ... -slab RANK WORLD IDFILE: this process owns planes [z0, z1) of the volume (WORLD processes, one GPU
// each).  Rank 0 makes the RCCL id and publishes it as IDFILE (written under a temporary name, then renamed); the other ranks
// wait for the file.  IDFILE "-" with WORLD 1 runs without a communicator.  Every rank reads the whole input, computes its
// owned planes (halos, the global top-fraction threshold and the overlapped votes are csrc/slab.hip's business) and returns
// them in `out` (nz = z1 - z0), which main() writes to this rank's own -out file; tools/join_slabs.py stacks the files.
// One rank's slab handle for `-slab RANK WORLD IDFILE`: rank 0 makes the RCCL id and publishes it as IDFILE (written under a
// temporary name, then renamed; removed again once every rank has joined); the other ranks wait for the file.  IDFILE "-"
// with WORLD 1 runs without a communicator.
visfd_hip_slab* open_slab(const Settings& s, int64_t nz, int ghost) {
  visfd_hip_ctx* ctx = hip_detail::context();
  unsigned char id[128];
  const bool with_id = !(s.slab_id_file == "-" && s.slab_world == 1);
  if (s.slab_id_file == "-" && s.slab_world > 1) throw VisfdErr("Error: -slab with more than one rank needs an id file.\n");
  if (with_id) {
    if (s.slab_rank == 0) {
      std::remove(s.slab_id_file.c_str());   // an id file left behind by an earlier run must not be picked up by this run's ranks
      hip_detail::check(visfd_hip_slab_unique_id(id));
      const string tmp = s.slab_id_file + ".tmp";
      { std::ofstream f(tmp.c_str(), std::ios::binary); f.write(reinterpret_cast<const char*>(id), 128);
        if (!f) throw VisfdErr("Error: unable to write \"" + tmp + "\".\n"); }
      if (std::rename(tmp.c_str(), s.slab_id_file.c_str()) != 0) throw VisfdErr("Error: unable to create \"" + s.slab_id_file + "\".\n");
    } else {
      // IDFILE must be unique per run.  Rank 0 removes it before it publishes a new id and again once the communicator is
      // up; a file older than the ten minutes a rank waits is taken to be a leftover and ignored.
      const std::time_t started = std::time(nullptr);
      bool got = false;
      for (int tries = 0; tries < 12000 && !got; tries++) {   // up to 10 minutes
        struct stat st;
        const bool fresh = ::stat(s.slab_id_file.c_str(), &st) == 0 && st.st_mtime + 600 >= started;
        std::ifstream f(s.slab_id_file.c_str(), std::ios::binary);
        if (fresh && f && f.read(reinterpret_cast<char*>(id), 128) && f.gcount() == 128) got = true;
        else std::this_thread::sleep_for(std::chrono::milliseconds(50));
      }
      if (!got) throw VisfdErr("Error: the id file \"" + s.slab_id_file + "\" did not appear (is rank 0 running?).\n");
    }
  }
  visfd_hip_slab* slab = nullptr;
  hip_detail::check(visfd_hip_slab_create_rccl(ctx, with_id ? id : nullptr, s.slab_rank, s.slab_world, nz, ghost, &slab));
  if (with_id && s.slab_rank == 0) std::remove(s.slab_id_file.c_str());   // every rank has joined: the id has served
  return slab;
}

// this rank's planes as an MRC file of their own: the input's header with nz, the cell's z extent and the z origin of the slab
void write_slab_part(const Settings& s, Mrc& tomo_in, Mrc& part, int64_t z0) {
  std::memcpy(part.raw_header, tomo_in.raw_header, 1024);
  float fw[256];
  std::memcpy(fw, part.raw_header, 1024);
  const float dz = tomo_in.cella[2] / (float)tomo_in.nz;
  part.cella[0] = tomo_in.cella[0]; part.cella[1] = tomo_in.cella[1]; part.cella[2] = dz * (float)part.nz;
  fw[51] += dz * (float)z0;                                // MRC2014 origin z (word 52)
  std::memcpy(part.raw_header, fw, 1024);
  if (!s.out.empty()) {
    cerr << "writing this slab's planes (in 32-bit float mode)\n";
    part.write(s.out, part);
  }
}

// -gauss ... -slab: this rank filters its owned planes (the ghost planes come from the neighbours; the normaliser follows
// global plane indices, so only the true faces of the volume are borders) and returns them in `out`.
void handle_gauss_slab(const Settings& s, Mrc& tomo_in, Mrc& out, float ratio, int64_t* z0_out) {
  if (!s.mask.empty()) throw VisfdErr("Error: -slab does not combine with -mask.\n");
  int hw[3];
  hip_detail::check(visfd_hip_gauss_halfwidths(s.width_a, ratio, hw));
  visfd_hip_slab* slab = open_slab(s, tomo_in.nz, hw[2]);
  int64_t lay[7];
  hip_detail::check(visfd_hip_slab_layout(slab, lay));
  const int64_t z0 = lay[0], z1 = lay[1];
  cerr << "slab " << s.slab_rank << " of " << s.slab_world << ": planes [" << z0 << ", " << z1 << "), ghost depth " << hw[2] << "\n";
  out.alloc(tomo_in.nx, tomo_in.ny, (int)(z1 - z0));
  const size_t plane = (size_t)tomo_in.nx * tomo_in.ny;
  float A = 0;
  const int rc = visfd_hip_apply_gauss_slab(slab, tomo_in.data() + (size_t)z0 * plane, tomo_in.nx, tomo_in.ny, s.width_a, hw,
                                            s.normalize ? 1 : 0, out.data(), &A);
  visfd_hip_slab_destroy(slab);
  hip_detail::check(rc);
  cerr << "  ... where  A = " << A << "\n";
  *z0_out = z0;
}

// -blob ... -slab: the blobs of this rank's owned planes (absolute score thresholds only: ratios need the global best score).
// Rows come back with GLOBAL z; every rank writes its own list files, tools/join_slabs.py merges them.
void handle_blob_slab(const Settings& s, Mrc& tomo_in, float ratio, vector<visfd_hip_blob>* mins, vector<visfd_hip_blob>* maxs) {
  if (!s.mask.empty()) throw VisfdErr("Error: -slab does not combine with -mask.\n");
  for (int d = 0; d < 3; d++)
    if (s.blob_aspect_ratio[d] != 1.0f) throw VisfdErr("Error: -slab runs isotropic blob detection only (no -blob-aspect-ratio).\n");
  vector<float> sig(s.blob_diameters.size());
  hip_detail::check(visfd_hip_blob_diameters_to_sigmas(s.blob_diameters.data(), (int)sig.size(), sig.data()));
  float smax = 0;
  for (size_t i = 0; i < sig.size(); i++) smax = std::max(smax, sig[i]);
  const int ghost = (int)std::floor(ratio * (double)smax * (1.0 + 0.5 * s.delta)) + 1;
  visfd_hip_slab* slab = open_slab(s, tomo_in.nz, ghost);
  int64_t lay[7];
  hip_detail::check(visfd_hip_slab_layout(slab, lay));
  const int64_t z0 = lay[0];
  cerr << "slab " << s.slab_rank << " of " << s.slab_world << ": planes [" << z0 << ", " << lay[1] << "), ghost depth " << ghost << "\n";
  const size_t plane = (size_t)tomo_in.nx * tomo_in.ny;
  int64_t cap = 1 << 16, nmin = 0, nmax = 0;
  int rc;
  for (;;) {
    mins->resize((size_t)cap);
    maxs->resize((size_t)cap);
    rc = visfd_hip_blob_dog_slab(slab, tomo_in.data() + (size_t)z0 * plane, tomo_in.nx, tomo_in.ny, sig.data(), (int)sig.size(), s.delta,
                                 ratio, s.score_upper, s.score_lower, mins->data(), cap, &nmin, maxs->data(), cap, &nmax);
    if (rc != VISFD_HIP_ECAPACITY) break;
    cap = std::max(std::max(nmin, nmax), cap) + 16;
  }
  visfd_hip_slab_destroy(slab);
  hip_detail::check(rc);
  mins->resize((size_t)nmin);
  maxs->resize((size_t)nmax);
}

void handle_membrane_slab(const Settings& s, Mrc& tomo_in, Mrc& out, float ratio, int order, int64_t* z0_out) {
  if (!(s.tv_sigma > 0)) throw VisfdErr("Error: -slab needs -tv (tensor voting).\n");
  if (!s.hessian_thr_is_fraction) throw VisfdErr("Error: -slab needs the fractional threshold (-tv-best), not -detection-threshold.\n");
  if (!s.mask.empty() || !s.load_base.empty() || !s.save_base.empty() || s.cluster_connected_voxels)
    throw VisfdErr("Error: -slab runs the plain -membrane ... -tv stage only (no -mask, -save/-load-progress, -connect).\n");
  int h_tv = 0;
  hip_detail::check(visfd_hip_tv_tables(s.tv_sigma, s.tv_truncate, &h_tv, nullptr, nullptr));
  const float sigma_bg = s.width_b[0] > 0.0f ? s.width_b[0] : 0.0f;
  const int ghost = std::max(std::max(h_tv, (int)std::floor(s.width_a[0] * ratio) + 1), (int)std::floor(sigma_bg * ratio));
  visfd_hip_slab* slab = open_slab(s, tomo_in.nz, ghost);
  int64_t lay[7];
  hip_detail::check(visfd_hip_slab_layout(slab, lay));
  const int64_t z0 = lay[0], z1 = lay[1];
  cerr << "slab " << s.slab_rank << " of " << s.slab_world << ": planes [" << z0 << ", " << z1 << "), ghost depth " << ghost << "\n";
  out.alloc(tomo_in.nx, tomo_in.ny, (int)(z1 - z0));
  float thr = 0;
  const size_t plane = (size_t)tomo_in.nx * tomo_in.ny;
  const int rc = visfd_hip_membrane_detect_slab_bg(slab, tomo_in.data() + (size_t)z0 * plane, tomo_in.nx, tomo_in.ny, s.width_a[0], ratio,
                                                   order, s.hessian_thr, s.tv_sigma, s.tv_exponent, s.tv_truncate, sigma_bg,
                                                   s.normalize ? 1 : 0, out.data(), nullptr, &thr);
  visfd_hip_slab_destroy(slab);
  hip_detail::check(rc);
  cerr << "  (saliency threshold = " << thr << ")\n";
  *z0_out = z0;
}

int main(int argc, char** argv) {
  try {
    cerr << "filter_mrc (visfd-mi355x, hot path on libvisfd_hip ABI " << visfd_hip_abi_version() << ")\n";
    Settings s = parse(argc, argv);
    Mrc tomo_in, mask, tomo_out;
    tomo_in.read(s.in);
    if (!s.mask.empty()) {
      mask.read(s.mask);
      if (mask.nx != tomo_in.nx || mask.ny != tomo_in.ny || mask.nz != tomo_in.nz)
        throw VisfdErr("Error: The size of the mask image does not match the size of the input image.\n");
    }
    int size[3] = {tomo_in.nx, tomo_in.ny, tomo_in.nz};
    float vw[3];
    if (s.voxel_width > 0) vw[0] = vw[1] = vw[2] = s.voxel_width;
    else {
      vw[0] = tomo_in.cella[0] / size[0];  // handlers.cpp:2429-2475: inferred from the header
      vw[1] = vw[2] = vw[0];
      if (!(vw[0] > 0)) vw[0] = vw[1] = vw[2] = 1.0f;
    }
    // ---- binning (filter_mrc.cpp:118-209): explicit (-bin N) or automatic for wide features ----
    const int size_orig[3] = {size[0], size[1], size[2]};
    const float cella_orig[3] = {tomo_in.cella[0], tomo_in.cella[1], tomo_in.cella[2]};
    int bin = s.bin;
    if (bin == 0) {
      bin = 1;
      if (s.tv_sigma > 0 && s.width_a[0] > 1.8 * vw[0])
        bin = (int)std::ceil(s.width_a[0] / (1.8 * vw[0]));
      else if (!(s.tv_sigma > 0) && !s.blob_diameters.empty() && s.blob_diameters[0] > 15.0 * vw[0])
        bin = (int)std::ceil(s.blob_diameters[0] / (15.0 * vw[0]));
      if (bin > 1)
        cerr << "--- WARNING: this would be very slow unless binning is used.\n"
                "--- BINNING THE IMAGE BY A FACTOR OF " << bin << "\n"
                "---           To prevent this, use the \"-bin 1\" argument.\n";
    }
    if (bin > 1) {
      const double w0 = s.voxel_width > 0 ? (double)s.voxel_width : (double)(tomo_in.cella[0] / tomo_in.nx);
      const double wb = w0 * bin;                      // handlers.cpp:2372-2385
      bin_image(tomo_in, bin, wb);
      if (mask.loaded) bin_image(mask, bin, wb);
      size[0] = tomo_in.nx; size[1] = tomo_in.ny; size[2] = tomo_in.nz;
      if (s.voxel_width > 0) vw[0] = vw[1] = vw[2] = s.voxel_width * bin;           // handlers.cpp:2445-2460
      else for (int d = 0; d < 3; d++) vw[d] = tomo_in.cella[d] / size[d];
    }
    cerr << "voxel width = " << vw[0] << "\n";
    for (size_t k = 0; k < s.must_link_crds.size(); k++)      // filter_mrc.cpp:372-379: physical units -> voxels, or
      s.must_link_crds[k] /= s.must_link_in_voxels ? (float)bin : vw[k % 3];   // voxels of the unbinned image -> binned
    for (int d = 0; d < 3; d++) { s.width_a[d] /= vw[d]; s.width_b[d] /= vw[d]; s.log_width[d] /= vw[d]; s.template_background_radius[d] /= vw[d]; }
    s.tv_sigma /= vw[0];
    for (size_t k = 0; k < s.blob_diameters.size(); k++) s.blob_diameters[k] /= vw[0];

    tomo_out.alloc(size[0], size[1], size[2]);
    std::memcpy(tomo_out.data(), tomo_in.data(), tomo_in.nvox() * 4);   // filter_mrc.cpp:398
    float const* const* const* M = mask.loaded ? mask.a : nullptr;
    const float ratio = ratio_of(s);

    if (s.slab_world > 0 && bin > 1) throw VisfdErr("Error: -slab does not combine with binning (use -bin 1).\n");
    if (s.slab_world > 0 && s.type != Settings::GAUSS && s.type != Settings::BLOB && s.type != Settings::SURFACE_RIDGE)
      throw VisfdErr("Error: -slab runs with -gauss, -blob and -membrane ... -tv.\n");
    if (s.type == Settings::GAUSS && s.slab_world > 0) {
      cerr << "filter_type = Gaussian (Z-slab mode)\n";
      Mrc part;
      int64_t z0 = 0;
      handle_gauss_slab(s, tomo_in, part, ratio, &z0);
      write_slab_part(s, tomo_in, part, z0);
      return 0;
    } else if (s.type == Settings::GAUSS) {
      cerr << "filter_type = Gaussian\n";
      const float A = ApplyGauss(size, tomo_in.a, tomo_out.a, M, s.width_a, s.truncate_ratio, s.truncate_threshold,
                                 s.normalize, &cerr);
      cerr << " Filter Used: A discrete Gaussian kernel, approximately equal to\n"
              " h(x,y,z)   ≈ A*exp(-0.5*((x/σ_x)^2 + (y/σ_y)^2 + (z/σ_z)^2))\n"
              " ... where  A = " << A << "\n";
    } else if (s.type == Settings::LOCAL_FLUCTUATIONS) {
      // HandleLocalFluctuations, handlers.cpp:1254-1271
      LocalFluctuationsByRadius(size, tomo_in.a, tomo_out.a, M, s.template_background_radius,
                                s.template_background_exponent, s.truncate_ratio, s.truncate_threshold, s.normalize, &cerr);
    } else if (s.type == Settings::DOG) {
      cerr << "filter_type = Difference of Gaussians (DoG)\n";
      // bin/filter_mrc/filter3d_variants.hpp:542-597: each Gaussian has its own window
      Mrc tmp;
      tmp.alloc(size[0], size[1], size[2]);
      const float A = ApplyGauss(size, tomo_in.a, tomo_out.a, M, s.width_a, s.truncate_ratio, s.truncate_threshold, true);
      const float B = ApplyGauss(size, tomo_in.a, tmp.a, M, s.width_b, s.truncate_ratio, s.truncate_threshold, true);
      float* o = tomo_out.data();
      const float* t = tmp.data();
      for (size_t i = 0; i < tomo_out.nvox(); i++) o[i] -= t[i];
      cerr << "  ... where      A = " << A << "\n                 B = " << B << "\n";
    } else if (s.type == Settings::LOG) {
      cerr << "filter_type = Laplacian of Gaussians (LoG)\n";
      float A = 0, B = 0;
      ApplyLog(size, tomo_in.a, tomo_out.a, M, s.log_width, s.delta, ratio, &A, &B, &cerr);
      cerr << "  ... where      A = " << A << "\n                 B = " << B << "\n";
    } else if (s.type == Settings::BLOB) {
      vector<std::array<float, 3> > cmin, cmax;
      vector<float> dmin, dmax, smin, smax;
      string slab_suffix;
      if (s.slab_world > 0) {
        // this rank's blobs (global z); with more than one rank every rank writes "<file>.slab<RANK>" (tools/join_slabs.py)
        vector<visfd_hip_blob> bl[2];
        handle_blob_slab(s, tomo_in, ratio, &bl[0], &bl[1]);
        for (int side = 0; side < 2; side++) {
          vector<std::array<float, 3> >& c = side ? cmax : cmin;
          vector<float>& dia = side ? dmax : dmin;
          vector<float>& sc = side ? smax : smin;
          vector<float> sg(bl[side].size());
          c.resize(sg.size()); dia.resize(sg.size()); sc.resize(sg.size());
          for (size_t i = 0; i < sg.size(); i++) {
            c[i][0] = (float)bl[side][i].ix; c[i][1] = (float)bl[side][i].iy; c[i][2] = (float)bl[side][i].iz;
            sg[i] = bl[side][i].sigma; sc[i] = bl[side][i].score;
          }
          if (!sg.empty()) hip_detail::check(visfd_hip_blob_sigmas_to_diameters(sg.data(), (int)sg.size(), dia.data()));
        }
        if (s.slab_world > 1) { std::ostringstream o; o << ".slab" << s.slab_rank; slab_suffix = o.str(); }
      } else
      BlobDogD(size, tomo_in.a, M, s.blob_diameters, &cmin, &cmax, &dmin, &dmax, &smin, &smax, s.blob_aspect_ratio, s.delta, ratio,
               s.score_upper, s.score_lower, false, &cerr);
      // physical units + sort by score (handlers.cpp:853-909), ties keep list order
      for (int side = 0; side < 2; side++) {
        const string fname = (side ? s.blob_max_file : s.blob_min_file).empty() ? string() : (side ? s.blob_max_file : s.blob_min_file) + slab_suffix;
        if (fname.empty()) continue;
        vector<std::array<float, 3> >& c = side ? cmax : cmin;
        vector<float>& dia = side ? dmax : dmin;
        vector<float>& sc = side ? smax : smin;
        vector<size_t> idx(c.size());
        for (size_t i = 0; i < idx.size(); i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](size_t p, size_t q) { return side ? sc[p] > sc[q] : sc[p] < sc[q]; });
        std::ofstream out(fname.c_str());
        if (!out) throw VisfdErr("Error: unable to open \"" + fname + "\" for writing.\n");
        for (size_t k = 0; k < idx.size(); k++) {
          const size_t i = idx[k];
          out << c[i][0] * vw[0] << " " << c[i][1] * vw[1] << " " << c[i][2] * vw[2] << " " << dia[i] * vw[0] << " "
              << sc[i] << "\n";
        }
      }
    } else if (s.type == Settings::BLOB_NONMAX) {
      handle_blob_nonmax(s, vw, M, size);
    } else if (s.type == Settings::SURFACE_RIDGE) {
      cerr << "filter_type = surface ridge detector\n";
      const int order = s.ridges_are_maxima ? VISFD_HIP_INCREASING_EIVALS : VISFD_HIP_DECREASING_EIVALS;  // handlers.cpp:1524-1535
      const size_t n = tomo_in.nvox();
      const bool want_tensor = s.tv_sigma > 0 && (!s.save_base.empty() || s.cluster_connected_voxels || !s.load_base.empty());
      vector<float> tensor(want_tensor ? 6 * n : 0);
      const float* mptr = mask.loaded ? mask.data() : nullptr;
      if (s.slab_world > 0) {
        Mrc part;
        int64_t z0 = 0;
        handle_membrane_slab(s, tomo_in, part, ratio, order, &z0);
        write_slab_part(s, tomo_in, part, z0);
        return 0;
      }
      if (s.load_base.empty()) {
        float thr = 0;
        hip_detail::check(visfd_hip_membrane_detect_bg(
            hip_detail::context(), tomo_in.data(), mptr, size[0], size[1], size[2],
            s.width_a[0], ratio, order, s.hessian_thr_is_fraction ? s.hessian_thr : -1.0f, s.hessian_thr, s.tv_sigma,
            s.tv_exponent, s.tv_truncate, s.width_b[0] > 0.0f ? s.width_b[0] : 0.0f, s.normalize ? 1 : 0, tomo_out.data(),
            tensor.empty() ? nullptr : tensor.data(), nullptr, &thr));
        cerr << "  (saliency threshold = " << thr << ")\n";
      } else {
        // handlers.cpp:1840-1862: the vote tensors come from "<base>_tensor_<d>.rec" (written by -save-progress)
        for (int c = 0; c < 6; c++) {
          std::ostringstream name;
          name << s.load_base << "_tensor_" << c << ".rec";
          cerr << "loading \"" << name.str() << "\"\n";
          Mrc t;
          t.read(name.str());
          if (t.nx != size[0] || t.ny != size[1] || t.nz != size[2])
            throw VisfdErr("Error: \"" + name.str() + "\" does not have the size of the (binned) input image.\n");
          const float* p = t.data();
          for (size_t i = 0; i < n; i++)
            if (!mptr || mptr[i] != 0.0f) tensor[6 * i + c] = p[i];
        }
        hip_detail::check(visfd_hip_tensor_saliency_host(tensor.data(), mptr, (int64_t)n, order, tomo_out.data()));
        if (s.width_b[0] > 0.0f) {   // the peak-height factor of the post-vote loop, handlers.cpp:1577-1592,1883-1887
          Mrc bgv;
          bgv.alloc(size[0], size[1], size[2]);
          const float sb[3] = {s.width_b[0], s.width_b[0], s.width_b[0]};
          const int hb = (int)std::floor(s.width_b[0] * ratio);
          const int hwb[3] = {hb, hb, hb};
          hip_detail::check(visfd_hip_apply_gauss(hip_detail::context(), tomo_in.data(), bgv.data(), mptr, size[0], size[1], size[2], sb, hwb,
                                                  s.normalize ? 1 : 0, nullptr));
          const float* img = tomo_in.data();
          const float* bg = bgv.data();
          float* o = tomo_out.data();
          for (size_t i = 0; i < n; i++)
            if (!mptr || mptr[i] != 0.0f) o[i] *= img[i] - bg[i];
        }
      }
      if (!tensor.empty() && !s.save_base.empty()) {
        Mrc t;
        t.alloc(size[0], size[1], size[2]);
        std::memcpy(t.raw_header, tomo_in.raw_header, 1024);
        for (int d = 0; d < 3; d++) t.cella[d] = tomo_in.cella[d];
        // (the reference starts each tensor file from a copy of tomo_out: masked voxels keep its values)
        for (int c = 0; c < 6; c++) {
          float* o = t.data();
          const float* base = tomo_out.data();
          for (size_t i = 0; i < n; i++) o[i] = (!mptr || mptr[i] != 0.0f) ? tensor[6 * i + c] : base[i];
          std::ostringstream name;
          name << s.save_base << "_tensor_" << c << ".rec";
          cerr << "writing \"" << name.str() << "\"\n";
          t.write(name.str(), tomo_in);
        }
      }
      if (s.cluster_connected_voxels) {
        // handlers.cpp:1925-2035.  Saliency and directions are recomputed on the host in the reference's own
        // arithmetic (the flood order and the angle thresholds act on them); the vote tensors are exact already.
        hip_detail::check(visfd_hip_tensor_saliency_host(tensor.data(), mptr, (int64_t)n, order, tomo_out.data()));
        vector<float> direction(3 * n, 0.0f);
        hip_detail::check(visfd_hip_principal_directions_host(tensor.data(), mptr, (int64_t)n, order, direction.data()));
        vector<int64_t> labels(n);
        int64_t n_clusters = 0;
        hip_detail::check(visfd_hip_label_connected_ex(
            tomo_out.data(), labels.data(), mptr, size[0], size[1], size[2], s.connect_threshold_saliency,
            direction.data(), s.connect_threshold_vector_saliency, s.connect_threshold_vector_neighbor, 0, tensor.data(),
            s.connect_threshold_tensor_saliency, s.connect_threshold_tensor_neighbor, 1, 1, -1, 1, 1, 1, &n_clusters,
            nullptr, nullptr, nullptr, 0, nullptr, s.must_link_crds.empty() ? nullptr : s.must_link_crds.data(),
            s.must_link_group_sizes.empty() ? nullptr : s.must_link_group_sizes.data(),
            (int64_t)s.must_link_group_sizes.size(), s.must_link_directions.empty() ? nullptr : s.must_link_directions.data()));
        cerr << "Number of clusters found: " << n_clusters << "\n";
        int64_t max_label = labels[0];
        for (size_t i = 0; i < n; i++)
          if (!mptr || mptr[i] != 0.0f) max_label = std::max(max_label, labels[i]);
        vector<float> saliency;
        if (!s.out_normals_file.empty()) saliency.assign(tomo_out.data(), tomo_out.data() + n);   // handlers.cpp:1929-1934
        float* o = tomo_out.data();
        for (size_t i = 0; i < n; i++) {
          o[i] = (float)labels[i];
          if (labels[i] == -1) o[i] = s.undefined_voxels_are_max ? (float)(max_label + 1) : s.undefined_voxel_brightness;
        }
        if (!s.out_normals_file.empty()) {   // handlers.cpp:2039-2309
          float maxd = s.max_distance_to_feature;                     // filter_mrc.cpp:301-307
          if (maxd < 0.0f) maxd /= -vw[0];
          else maxd /= (float)bin;
          int64_t np = 0;
          hip_detail::check(visfd_hip_surface_points(saliency.data(), o, direction.data(), mptr, size[0], size[1], size[2],
                                                     s.select_cluster, vw, s.surface_normal_curve_ds, s.surface_find_ridge ? 1 : 0,
                                                     maxd, nullptr, nullptr, 0, &np));
          vector<float> crds(3 * (size_t)np + 3), norms(3 * (size_t)np + 3);
          hip_detail::check(visfd_hip_surface_points(saliency.data(), o, direction.data(), mptr, size[0], size[1], size[2],
                                                     s.select_cluster, vw, s.surface_normal_curve_ds, s.surface_find_ridge ? 1 : 0,
                                                     maxd, crds.data(), norms.data(), np, &np));
          std::ofstream ply(s.out_normals_file.c_str());              // file_io.hpp:501-527
          if (!ply) throw VisfdErr("Error: unable to open \"" + s.out_normals_file + "\" for writing.\n");
          ply << "ply\nformat ascii 1.0\ncomment  created by visfd\nelement vertex " << np
              << "\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\n"
                 "property float nz\nend_header\n";
          for (int64_t k = 0; k < np; k++)
            ply << crds[3 * k] << " " << crds[3 * k + 1] << " " << crds[3 * k + 2] << " " << norms[3 * k] << " "
                << norms[3 * k + 1] << " " << norms[3 * k + 2] << "\n";
        }
      }
    }
    if (s.type == Settings::SURFACE_RIDGE && bin > 1 && !s.bin_explicit) {   // handlers.cpp:2315-2355
      tomo_out.loaded = true;
      std::memcpy(tomo_out.raw_header, tomo_in.raw_header, 1024);
      unbin_image(tomo_out, size_orig, cella_orig);
      unbin_image(tomo_in, size_orig, cella_orig);   // only its header/size is used below
      if (mask.loaded) unbin_image(mask, size_orig, cella_orig);
    }
    // filter_mrc.cpp:765-776: after everything else, voxels outside the mask take the "masked" brightness
    if (mask.loaded && s.type != Settings::BLOB && s.type != Settings::BLOB_NONMAX) {
      float* o = tomo_out.data();
      const float* mp = mask.data();
      for (size_t i = 0; i < tomo_out.nvox(); i++)
        if (mp[i] == 0.0f) o[i] = s.masked_voxel_brightness;
    }
    if (!s.out.empty()) {
      cerr << "writing tomogram (in 32-bit float mode)\n";
      tomo_out.write(s.out, tomo_in);
    }
  } catch (std::exception& e) {
    cerr << "\n" << e.what() << std::endl;
    return 1;
  }
  return 0;
}
