// C++ host mirror smoke: the reference's README example (README.md:92-150) written against
// include/hamilton.hpp.  Without arguments: build the System (tape -> hiprtc) and print the
// generated source (works without a GPU).  With "run": evaluate on the GPU and print numbers.
#include <cstdio>
#include <cstring>

#include "hamilton.hpp"

using hamilton::Var;

int main(int argc, char** argv) {
  const double m1 = 1.0, m2 = 1.0;
  // doublePendulum m1 m2 (app/Examples.hs:75-94)
  hamilton::System s = hamilton::mkSystemP(
      {m1, m1, m2, m2}, 2,
      [](const std::vector<Var>& q) {
        using hamilton::sin; using hamilton::cos;
        const Var &t1 = q[0], &t2 = q[1];
        return std::vector<Var>{sin(t1), 1 - cos(t1), sin(t1) + sin(t2) / 2, 1 - cos(t1) - cos(t2) / 2};
      },
      [=](const std::vector<Var>& x) { return 5 * (m1 * x[1] + m2 * x[3]); });
  if (argc < 2 || std::strcmp(argv[1], "run") != 0) {
    std::printf("%s", s.source().c_str());
    return 0;
  }
  hamilton::Config c0 = hamilton::Cfg({M_PI / 2, 0.0}, {0.0, 0.0});          // seInit (Examples.hs:94)
  hamilton::Phase p0 = hamilton::toPhase(s, c0);
  auto d = hamilton::hamEqs(s, p0);
  std::printf("hamEqs dq = %.17g %.17g dp = %.17g %.17g\n", d.first[0], d.first[1], d.second[0], d.second[1]);
  hamilton::Phase p1 = hamilton::stepHam(0.01, s, p0);
  std::printf("stepHam q = %.17g %.17g p = %.17g %.17g\n", p1.positions[0], p1.positions[1], p1.momenta[0], p1.momenta[1]);
  auto rows = hamilton::evolveHam(s, p0, {0.0, 0.01, 0.02});
  std::printf("evolveHam rows = %zu q1 = %.17g\n", rows.size(), rows[1].positions[0]);
  std::printf("hamiltonian = %.17g\n", hamilton::hamiltonian(s, p1)[0]);
  auto c1 = hamilton::stepHamC(0.01, s, c0);
  std::printf("stepHamC qd = %.17g %.17g\n", c1.velocities[0], c1.velocities[1]);
  return 0;
}
