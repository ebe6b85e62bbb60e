// Compiled and run by tests/test_abi.py: the C++11 shim (include/visfd_hip.hpp) must compile with -Wall under g++ and
// its host-side entry points (no GPU needed) must work through the reference's own signatures.
#include <cmath>
#include <cstdio>

#include "visfd_hip.hpp"

using namespace visfd;

int main() {
  // LabelConnected (connect.hpp:168-197): two separate bright boxes -> two clusters, the larger one is label 1
  int size[3] = {12, 10, 9};
  float*** sal = Alloc3D<float>(size);
  ptrdiff_t*** lab = Alloc3D<ptrdiff_t>(size);
  for (int z = 0; z < 9; z++)
    for (int y = 0; y < 10; y++)
      for (int x = 0; x < 12; x++)
        sal[z][y][x] = (x > 2 && x < 6 && y > 2 && y < 7) ? 5.0f + 0.01f * (x + y + z) : ((x > 8) ? 3.0f + 0.02f * z : 0.0f);
  std::vector<std::array<float, 3> > cm;
  std::vector<float> cs, csal;
  const float ninf = -std::numeric_limits<float>::infinity();
  size_t n = LabelConnected(size, sal, lab, nullptr, 1.0f, nullptr, ninf, ninf, true, nullptr, ninf, ninf, true, 1,
                            (ptrdiff_t)-1, &cm, &cs, &csal);
  long c1 = 0, c2 = 0, cu = 0;
  for (int z = 0; z < 9; z++)
    for (int y = 0; y < 10; y++)
      for (int x = 0; x < 12; x++) {
        if (lab[z][y][x] == 1) c1++;
        else if (lab[z][y][x] == 2) c2++;
        else if (lab[z][y][x] == -1) cu++;
      }
  std::printf("clusters %zu sizes %ld %ld undefined %ld\n", n, c1, c2, cu);
  if (!(n == 2 && c1 == 270 && c2 == 108 && cu == 702 && cm.size() == 2)) return 1;

  // blob list post-processing (feature.hpp:519-913)
  std::vector<std::array<float, 3> > crds;
  std::vector<float> diam, score;
  const float pts[4][5] = {{10, 10, 10, 6, -5}, {11, 10, 10, 6, -9}, {30, 30, 30, 4, -2}, {30, 31, 30, 4, 1}};
  for (int i = 0; i < 4; i++) {
    std::array<float, 3> c = {{pts[i][0], pts[i][1], pts[i][2]}};
    crds.push_back(c); diam.push_back(pts[i][3]); score.push_back(pts[i][4]);
  }
  DiscardOverlappingBlobs(crds, diam, score, 1.0f);
  std::printf("blobs kept %zu best score %g overlap %g\n", crds.size(), score[0], CalcSphereOverlap(1.0f, 2.0f, 3.0f));
  if (!(crds.size() == 2 && score[0] == -9.0f && score[1] == -2.0f)) return 2;
  std::vector<size_t> perm;
  SortBlobs(crds, diam, score, SORT_DECREASING, false, &perm);   // descending by signed score
  if (!(score[0] == -2.0f && perm.size() == 2 && perm[0] == 1)) return 3;
  // per-voxel eigen helpers (eigen3_simple.hpp:271, :392; feature.hpp:1526-1612): a diagonal matrix has its
  // entries as eigenvalues, decreasing order puts the largest first, and the eigenvector rows are orthonormal
  const float m6[6] = {2.0f, 5.0f, -1.0f, 0.0f, 0.0f, 0.0f};
  float d6[6], ev[3], E[3][3];
  selfadjoint_eigen3::DiagonalizeFlatSym3(m6, d6, selfadjoint_eigen3::DECREASING_EIVALS);
  selfadjoint_eigen3::ConvertFlatSym2Evects3(m6, ev, E, selfadjoint_eigen3::DECREASING_EIVALS);
  std::printf("eigen %g %g %g planar %g stick %g\n", d6[0], d6[1], d6[2], ScoreHessianPlanar(d6), ScoreTensorPlanar(d6));
  if (!(d6[0] == 5.0f && d6[1] == 2.0f && d6[2] == -1.0f && ev[0] == 5.0f)) return 4;
  if (!(ScoreHessianPlanar(d6) == 441.0 && ScoreTensorPlanar(d6) == 3.0 && ScoreTensorLinear(d6) == 9.0)) return 5;
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      float dot = E[a][0] * E[b][0] + E[a][1] * E[b][1] + E[a][2] * E[b][2];
      if (std::fabs(dot - (a == b ? 1.0f : 0.0f)) > 1e-5f) return 6;
    }
  Dealloc3D(sal);
  Dealloc3D(lab);
  return 0;
}
