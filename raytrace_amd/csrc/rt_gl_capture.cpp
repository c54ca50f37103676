// rt_gl_capture.cpp — host images of the uniform buffers the scene headers write.
//
// The reference's scene classes serialise themselves with
//   glBindBuffer(GL_UNIFORM_BUFFER, buf); glBufferSubData(GL_UNIFORM_BUFFER, off, n, p);
//   glBindBuffer(GL_UNIFORM_BUFFER, 0);
// (sphere.h:24-34, material.h:30-44), where glBindBuffer / glBufferSubData are
// macros for the global function pointers glad_glBindBuffer /
// glad_glBufferSubData (include/glad/glad.h:1812-1814, 1827-1829).  Pointing
// those two pointers at the functions below turns every add() into a write
// into a byte vector per buffer id; rt_scene_upload then takes the bytes.
// State is per thread, like a GL context is current per thread.

#include <cstring>
#include <map>
#include <vector>

#include "../../include/raytrace_amd.h"

namespace {
struct capture_state {
  std::map<uint32_t, uint32_t> bound;                  // target -> buffer id
  std::map<uint32_t, std::vector<uint8_t>> buffers;    // buffer id -> bytes
  uint64_t dropped = 0;
};
thread_local capture_state g_cap;
}  // namespace

extern "C" {

void rt_gl_bind_buffer(uint32_t target, uint32_t buffer) { g_cap.bound[target] = buffer; }

void rt_gl_buffer_data(uint32_t target, intptr_t size, const void* data, uint32_t usage) {
  (void)usage;
  auto it = g_cap.bound.find(target);
  if (it == g_cap.bound.end() || it->second == 0 || size < 0) {
    g_cap.dropped++;
    return;
  }
  std::vector<uint8_t>& b = g_cap.buffers[it->second];
  b.assign((size_t)size, 0);
  if (data && size > 0) std::memcpy(b.data(), data, (size_t)size);
}

void rt_gl_buffer_sub_data(uint32_t target, intptr_t offset, intptr_t size, const void* data) {
  auto it = g_cap.bound.find(target);
  if (it == g_cap.bound.end() || it->second == 0 || offset < 0 || size < 0 || (size > 0 && !data)) {
    g_cap.dropped++;
    return;
  }
  std::vector<uint8_t>& b = g_cap.buffers[it->second];
  const size_t end = (size_t)offset + (size_t)size;
  if (b.size() < end) b.resize(end, 0);
  if (size > 0) std::memcpy(b.data() + offset, data, (size_t)size);
}

const void* rt_gl_buffer_bytes(uint32_t buffer, size_t* size) {
  auto it = g_cap.buffers.find(buffer);
  if (it == g_cap.buffers.end()) {
    if (size) *size = 0;
    return nullptr;
  }
  if (size) *size = it->second.size();
  return it->second.data();
}

void rt_gl_reset(void) {
  g_cap.bound.clear();
  g_cap.buffers.clear();
  g_cap.dropped = 0;
}

uint64_t rt_gl_dropped_writes(void) { return g_cap.dropped; }

}  // extern "C"
