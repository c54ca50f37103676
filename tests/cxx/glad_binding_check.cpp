// Compile-level check of the drop-in binding (container only: needs the reference tree).
// The reference's glad.h and hittable_list.h are included where they lie; src/glad.c is compiled where it lies and
// linked in, so glad_glBindBuffer / glad_glBufferSubData / glad_glBufferData are the reference's own globals
// (include/glad/glad.h:1812-1829, src/glad.c:285,319).  The library's capture functions are assigned to them
// WITHOUT casts — a signature drift would not compile — and a hittable written the way sphere::add is written
// (sphere.h:22-35: bind, glBufferSubData per field, unbind) is driven through the reference's hittable_list::add.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include <glad/glad.h>      // /root/reference/include/glad/glad.h
#include "hittable.h"       // /root/reference/hittable.h
#include "hittable_list.h"  // /root/reference/hittable_list.h
#include "raytrace_amd.h"

static_assert(std::is_same<PFNGLBINDBUFFERPROC, decltype(&rt_gl_bind_buffer)>::value, "rt_gl_bind_buffer is not a PFNGLBINDBUFFERPROC");
static_assert(std::is_same<PFNGLBUFFERSUBDATAPROC, decltype(&rt_gl_buffer_sub_data)>::value, "rt_gl_buffer_sub_data is not a PFNGLBUFFERSUBDATAPROC");
static_assert(std::is_same<PFNGLBUFFERDATAPROC, decltype(&rt_gl_buffer_data)>::value, "rt_gl_buffer_data is not a PFNGLBUFFERDATAPROC");

// The record sphere::add writes (sphere.h:24-34), through the gl* macros of glad.h, i.e. through the glad pointers.
struct record : public hittable {
  int mat;
  float radius;
  float origin[3];
  record(int m, float r, float x, float y, float z) : hittable{32}, mat(m), radius(r), origin{x, y, z} {}
  void add(unsigned int glBuffer, int offset) const override {
    glBindBuffer(GL_UNIFORM_BUFFER, glBuffer);
    glBufferSubData(GL_UNIFORM_BUFFER, offset, 4, &mat);
    glBindBuffer(GL_UNIFORM_BUFFER, 0);
    glBindBuffer(GL_UNIFORM_BUFFER, glBuffer);
    glBufferSubData(GL_UNIFORM_BUFFER, offset + 4, 4, &radius);
    glBindBuffer(GL_UNIFORM_BUFFER, 0);
    glBindBuffer(GL_UNIFORM_BUFFER, glBuffer);
    glBufferSubData(GL_UNIFORM_BUFFER, offset + 16, 12, origin);
    glBindBuffer(GL_UNIFORM_BUFFER, 0);
  }
};

int main() {
  glad_glBindBuffer = rt_gl_bind_buffer;          // no casts
  glad_glBufferSubData = rt_gl_buffer_sub_data;
  glad_glBufferData = rt_gl_buffer_data;
  const unsigned sphereUBO = 2;
  glBindBuffer(GL_UNIFORM_BUFFER, sphereUBO);     // raytrace.cpp:418-420
  glBufferData(GL_UNIFORM_BUFFER, 32768, NULL, GL_STATIC_DRAW);
  glBindBuffer(GL_UNIFORM_BUFFER, 0);
  hittable_list objects;
  record a(0, 100.0f, 0.0f, -100.5f, -1.0f), b(1, 0.5f, 0.0f, 0.0f, -1.2f), c(4, 0.4f, -1.0f, 0.0f, -1.0f);
  objects.add(sphereUBO, a);
  objects.add(sphereUBO, b);
  objects.add(sphereUBO, c);
  size_t size = 0;
  const unsigned char* bytes = static_cast<const unsigned char*>(rt_gl_buffer_bytes(sphereUBO, &size));
  if (objects.num != 3 || objects.offset != 96 || bytes == nullptr || size != 32768 || rt_gl_dropped_writes() != 0) {
    std::printf("FAIL num=%d offset=%d size=%zu dropped=%llu\n", objects.num, objects.offset, size, (unsigned long long)rt_gl_dropped_writes());
    return 1;
  }
  const record* want[3] = {&a, &b, &c};
  for (int i = 0; i < 3; i++) {
    int m;
    float r, o[3];
    std::memcpy(&m, bytes + 32 * i, 4);
    std::memcpy(&r, bytes + 32 * i + 4, 4);
    std::memcpy(o, bytes + 32 * i + 16, 12);
    if (m != want[i]->mat || r != want[i]->radius || std::memcmp(o, want[i]->origin, 12) != 0) {
      std::printf("FAIL record %d\n", i);
      return 1;
    }
  }
  std::printf("OK 3 records through the reference's glad pointers and hittable_list\n");
  return 0;
}
