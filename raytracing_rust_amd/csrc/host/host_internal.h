// Internals shared by the host scene layer's translation units.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../../include/rtx_host.h"

namespace rtxh {

// SplitMix64 stream for random scene content (box heights, sphere centres, Perlin tables):
//   state += 0x9E3779B97F4A7C15; z = state; z = (z ^ z>>30) * 0xBF58476D1CE4E5B9;
//   z = (z ^ z>>27) * 0x94D049BB133111EB; z ^= z>>31; u = (z >> 11) * 2^-53.
// range(a,b) = a + (b-a)*u (random.rs:16-19); int_range(a,b) = a + floor(u*(b-a+1)) (inclusive, random.rs:22-25).
struct HostRng {
  uint64_t state;
  explicit HostRng(uint64_t seed) : state(seed) {}
  double next_double();
  double range(double mn, double mx);
  int int_range(int mn, int mx);
};

void perlin_from_rng(HostRng& rng, rtx_perlin& P);
int add_noise_texture(rtxh_session* s, double scale, const rtx_perlin& P);
bool load_ppm(const char* path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h);

}  // namespace rtxh
