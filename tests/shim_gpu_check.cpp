// Compiled and run by tests/test_shim_gpu.py on the GPU box: a caller written against the reference's template API
// (float*** arrays from Alloc3D, CompactMultiChannelImage3D, std::vector result lists) runs through
// include/visfd_hip.hpp.  Inputs come from <dir>/in.bin, every result goes to <dir>/out.bin as (name, count, floats)
// records; the Python side compares them with the oracle and the goldens.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "visfd_hip.hpp"

using namespace visfd;

static FILE* g_out = nullptr;

static void put(const char* name, const std::vector<float>& v) {
  char tag[32];
  std::memset(tag, 0, sizeof tag);
  std::strncpy(tag, name, sizeof tag - 1);
  const long long n = (long long)v.size();
  fwrite(tag, 1, sizeof tag, g_out);
  fwrite(&n, sizeof n, 1, g_out);
  if (n) fwrite(v.data(), sizeof(float), (size_t)n, g_out);
}

static std::vector<float> take(FILE* f, size_t n) {
  std::vector<float> v(n);
  if (fread(v.data(), sizeof(float), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}

static float*** wrap(const std::vector<float>& flat, const int size[3]) {
  float*** a = Alloc3D<float>(size);
  std::memcpy(&a[0][0][0], flat.data(), sizeof(float) * flat.size());
  return a;
}

static std::vector<float> flat_of(float*** a, size_t n) { return std::vector<float>(&a[0][0][0], &a[0][0][0] + n); }

// tensors of a compact container as 6 floats per voxel, -7 where a voxel has no storage
static std::vector<float> flat_tensor(CompactMultiChannelImage3D<float>& t, const int size[3]) {
  std::vector<float> out;
  for (int iz = 0; iz < size[2]; iz++)
    for (int iy = 0; iy < size[1]; iy++)
      for (int ix = 0; ix < size[0]; ix++)
        for (int c = 0; c < 6; c++) out.push_back(t.aaaafI[iz][iy][ix] ? t.aaaafI[iz][iy][ix][c] : -7.0f);
  return out;
}

static std::vector<float> blob_rows(const std::vector<std::array<float, 3> >& c, const std::vector<float>& d,
                                    const std::vector<float>& s) {
  std::vector<float> out;
  for (size_t i = 0; i < c.size(); i++) {
    out.push_back(c[i][0]); out.push_back(c[i][1]); out.push_back(c[i][2]); out.push_back(d[i]); out.push_back(s[i]);
  }
  return out;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  FILE* f = fopen((dir + "/in.bin").c_str(), "rb");
  if (!f) return 2;
  int size[3];
  float tv_sigma = 0;
  if (fread(size, sizeof(int), 3, f) != 3 || fread(&tv_sigma, sizeof(float), 1, f) != 1) return 2;
  const size_t n = (size_t)size[0] * size[1] * size[2];
  std::vector<float> vsrc = take(f, n), vmask = take(f, n), vsal = take(f, n), vdir = take(f, 3 * n);
  fclose(f);
  g_out = fopen((dir + "/out.bin").c_str(), "wb");
  try {
    float*** src = wrap(vsrc, size);
    float*** mask = wrap(vmask, size);
    float*** sal = wrap(vsal, size);
    float*** dst = Alloc3D<float>(size);

    // ---- ApplySeparable with three different Gaussian filters, masked and normalised (filter3d.hpp:686-695)
    Filter1D<float, int> filt[3] = {GenFilterGauss1D(1.2f, 3), GenFilterGauss1D(1.5f, 4), GenFilterGauss1D(0.9f, 2)};
    const float A = ApplySeparable(size, src, dst, mask, filt, true);
    put("separable", flat_of(dst, n));
    put("separable_A", std::vector<float>(1, A));

    // ---- ApplyDog (filter3d.hpp:1338-1351)
    const float sa[3] = {1.0f, 1.1f, 1.2f}, sb[3] = {1.6f, 1.7f, 1.8f};
    const int hw[3] = {4, 4, 5};
    ApplyDog(size, src, dst, nullptr, sa, sb, hw);
    put("dog", flat_of(dst, n));

    // ---- BlobDog with sigmas and ratio thresholds (feature.hpp:53-77)
    {
      std::vector<float> sig;
      sig.push_back(1.0f); sig.push_back(1.3f); sig.push_back(1.7f); sig.push_back(2.2f);
      std::vector<std::array<float, 3> > cmin, cmax;
      std::vector<float> smin, smax, scmin, scmax;
      BlobDog(size, src, mask, sig, &cmin, &cmax, &smin, &smax, &scmin, &scmax, nullptr, 0.02f, 2.5f, 0.5f, 0.5f, true);
      put("blob_min", blob_rows(cmin, smin, scmin));
      put("blob_max", blob_rows(cmax, smax, scmax));
      // BlobDogNM: diameters, thresholds off, overlapping blobs discarded (feature_variants.hpp:393-503)
      std::vector<float> diam;
      diam.push_back(3.5f); diam.push_back(4.5f); diam.push_back(5.9f); diam.push_back(7.6f);
      std::vector<std::array<float, 3> > nmin, nmax;
      std::vector<float> dmin, dmax, nsmin, nsmax;
      BlobDogNM(size, src, nullptr, diam, &nmin, &nmax, &dmin, &dmax, &nsmin, &nsmax, nullptr, 0.02f, 2.5f, 0.9f, 0.9f,
                true, 1.0f, 1.0f, 1.0f);
      put("nm_min", blob_rows(nmin, dmin, nsmin));
      put("nm_max", blob_rows(nmax, dmax, nsmax));
    }

    // ---- CalcHessian into a compact container (feature.hpp:1203-1219, handlers.cpp:1547-1565)
    {
      std::array<float, 3>*** grad = Alloc3D<std::array<float, 3> >(size);
      for (size_t i = 0; i < n; i++) grad[0][0][i][0] = grad[0][0][i][1] = grad[0][0][i][2] = -7.0f;
      CompactMultiChannelImage3D<float> hess(6, size, mask);
      CalcHessian(size, src, grad, hess.aaaafI, mask, 1.4f, 2.5f);
      put("hessian", flat_tensor(hess, size));
      put("gradient", std::vector<float>(&grad[0][0][0][0], &grad[0][0][0][0] + 3 * n));
      CompactMultiChannelImage3D<float> copy(hess);   // deep copy: pointers into its own array
      put("hessian_copy", flat_tensor(copy, size));
      Dealloc3D(grad);
    }

    // ---- TV3D::TVDenseStick (feature.hpp:1711-1901) on array<float,3>*** normals and compact tensors
    {
      std::array<float, 3>*** v = Alloc3D<std::array<float, 3> >(size);
      std::memcpy(&v[0][0][0][0], vdir.data(), sizeof(float) * 3 * n);
      TV3D<float, int, std::array<float, 3>, float*> tv(tv_sigma, 4, std::sqrt(2.0f));
      const struct { const char* name; bool ms, md, norm, diag; } cases[] = {
          {"tv_plain", false, false, false, false},  {"tv_masked", true, true, false, false},
          {"tv_default_args", true, true, true, false},   // normalize = true is the reference's default argument
          {"tv_norm_dst_only", false, true, true, false}, {"tv_norm_no_dst", true, false, true, false},
          {"tv_diag", true, true, false, true}};
      for (size_t k = 0; k < sizeof cases / sizeof cases[0]; k++) {
        CompactMultiChannelImage3D<float> ten(6, size, cases[k].md ? mask : nullptr);
        if (std::strcmp(cases[k].name, "tv_default_args") == 0)
          tv.TVDenseStick(size, sal, v, ten.aaaafI, mask, mask);
        else
          tv.TVDenseStick(size, sal, v, ten.aaaafI, cases[k].ms ? mask : nullptr, cases[k].md ? mask : nullptr, false,
                          cases[k].norm, cases[k].diag);
        put(cases[k].name, flat_tensor(ten, size));
      }
      Dealloc3D(v);
    }
    Dealloc3D(src); Dealloc3D(mask); Dealloc3D(sal); Dealloc3D(dst);
  } catch (const VisfdErr& e) {
    fprintf(stderr, "VisfdErr: %s\n", e.what());
    return 1;
  }
  fclose(g_out);
  printf("shim gpu check ok\n");
  return 0;
}
