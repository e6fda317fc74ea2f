// C++ host mirror smoke: the reference's README example (README.md:92-150) written against
// include/hamilton.hpp.  Without arguments: build the System (tape -> hiprtc) and print the
// generated source (works without a GPU).  With "run": evaluate on the GPU and print numbers.
#include <cmath>
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
  // `iterate (stepHam 0.01)` (README.md:150) as one launch == the calls one by one, bit for bit
  hamilton::Phase it = p0;
  for (int k = 0; k < 6; ++k) it = hamilton::stepHam(0.01, s, it);
  std::vector<hamilton::Phase> frames;
  hamilton::Phase one = hamilton::iterateStepHam(0.01, 6, s, p0, 3, &frames);
  const bool same = std::memcmp(it.positions.data(), one.positions.data(), 16) == 0 && std::memcmp(it.momenta.data(), one.momenta.data(), 16) == 0 &&
                    frames.size() == 2 && std::memcmp(frames[1].positions.data(), one.positions.data(), 16) == 0;
  std::printf("iterateStepHam same = %d frames = %zu\n", same ? 1 : 0, frames.size());
  // the library's choices through the ABI (hamk_options), not the environment
  hamk_options o;
  hamk_options_init(&o);
  o.trig = HAMK_TRIG_DIRECT; o.rkf_body = HAMK_BODY_STAGE_LOOP; o.gsl_api = 1;
  hamilton::System s2 = hamilton::mkSystemP({m1, m1, m2, m2}, 2,
      [](const std::vector<Var>& q) {
        using hamilton::sin; using hamilton::cos;
        return std::vector<Var>{sin(q[0]), 1 - cos(q[0]), sin(q[0]) + sin(q[1]) / 2, 1 - cos(q[0]) - cos(q[1]) / 2};
      },
      [=](const std::vector<Var>& x) { return 5 * (m1 * x[1] + m2 * x[3]); }, &o);
  const hamk_options r = s2.options();
  hamilton::Phase p2 = hamilton::stepHam(0.01, s2, p0);
  std::printf("options trig = %d rkf_body = %d gsl_api = %d lanes = %d dq = %.3g\n", r.trig, r.rkf_body, r.gsl_api, r.lanes_per_trajectory,
              std::fabs(p2.positions[0] - p1.positions[0]));
  auto c1 = hamilton::stepHamC(0.01, s, c0);
  std::printf("stepHamC qd = %.17g %.17g\n", c1.velocities[0], c1.velocities[1]);
  return 0;
}
