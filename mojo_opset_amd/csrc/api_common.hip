// Error text, version, run-time switches and the last-launch note.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace mojo {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local char g_launch[160] = "";
static thread_local char g_history[1024] = "";          // notes since the last clear, '|'-separated, oldest dropped when full
static thread_local char g_history_out[1024] = "";
void note_launch(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_launch, sizeof(g_launch), fmt, ap);
  va_end(ap);
  const size_t have = strlen(g_history), add = strlen(g_launch);
  if (have + add + 2 > sizeof(g_history)) {              // drop the older half
    const char* cut = strchr(g_history + have / 2, '|');
    if (cut) memmove(g_history, cut + 1, strlen(cut + 1) + 1); else g_history[0] = '\0';
  }
  const size_t at = strlen(g_history);
  if (at + add + 2 <= sizeof(g_history)) {
    if (at) g_history[at] = '|';
    memcpy(g_history + at + (at ? 1 : 0), g_launch, add + 1);
  }
}

// ---- run-time switches (common.h: EnvSwitch / MOJO_SWITCH) -----------------------------------------------------------
std::atomic<uint32_t> g_env_generation{1};
static std::atomic<EnvSwitch*> g_switches{nullptr};

EnvSwitch::EnvSwitch(const char* n) : name(n) {
  EnvSwitch* head = g_switches.load(std::memory_order_relaxed);
  do {
    next = head;
  } while (!g_switches.compare_exchange_weak(head, this, std::memory_order_release, std::memory_order_relaxed));
}

uint64_t env_refresh(EnvSwitch& s) {
  const uint32_t gen = g_env_generation.load(std::memory_order_relaxed);
  const char* e = getenv(s.name);
  uint64_t st = static_cast<uint64_t>(gen) << 33;
  if (e && e[0] != '\0') {
    char* end = nullptr;
    const long long v = strtoll(e, &end, 10);
    if (end != e) st |= (uint64_t{1} << 32) | static_cast<uint32_t>(static_cast<int32_t>(v));
    else st |= (uint64_t{1} << 32) | static_cast<uint32_t>(switch_word(e));          // a word: its first four characters
  }
  s.state.store(st, std::memory_order_relaxed);
  return st;
}
}  // namespace mojo

// MOJO_SRC_HASH: sha256 (first 16 hex digits) over csrc/*.hip, csrc/*.h, csrc/experiments/*.h and include/*.h, given
// by csrc/build.py to THIS file only; the Python loader recomputes it from the tree beside the library and refuses a
// library built from other sources (backends/hip/lib.py).
#ifndef MOJO_SRC_HASH
#define MOJO_SRC_HASH "unstamped"
#endif
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
#define MOJO_BUILD_KIND "+experiments"
#else
#define MOJO_BUILD_KIND ""
#endif
extern "C" const char* mojo_hip_version(void) {
  return "mojo_hip 0.3.0 (gfx950" MOJO_BUILD_KIND ") src=" MOJO_SRC_HASH;
}
extern "C" const char* mojo_hip_last_error(void) { return mojo::g_err; }

extern "C" void mojo_hip_reload_env(void) {
  uint32_t g = mojo::g_env_generation.load(std::memory_order_relaxed) + 1;
  if ((g & 0x7fffffffu) == 0) g = 1;                       // 31 bits are kept per switch; 0 means "never read"
  mojo::g_env_generation.store(g & 0x7fffffffu, std::memory_order_relaxed);
}

extern "C" const char* mojo_hip_last_launch(void) { return mojo::g_launch; }

extern "C" const char* mojo_hip_launch_history(int clear) {
  memcpy(mojo::g_history_out, mojo::g_history, sizeof(mojo::g_history));
  if (clear) mojo::g_history[0] = '\0';
  return mojo::g_history_out;
}

// "NAME=value" (or "NAME=" when unset) of every switch a launcher has read so far in this process, space-separated, into
// buf (NUL-terminated, truncated to `capacity`); returns the number of switches.
extern "C" int64_t mojo_hip_switches(char* buf, int64_t capacity) {
  int64_t n = 0, pos = 0;
  if (buf && capacity > 0) buf[0] = '\0';
  const uint32_t gen = mojo::g_env_generation.load(std::memory_order_relaxed);
  for (mojo::EnvSwitch* s = mojo::g_switches.load(std::memory_order_acquire); s; s = s->next, ++n) {
    if (!buf || capacity <= 0) continue;
    uint64_t st = s->state.load(std::memory_order_relaxed);
    if (static_cast<uint32_t>(st >> 33) != gen) st = mojo::env_refresh(*s);
    char item[96];
    if ((st >> 32) & 1) snprintf(item, sizeof(item), "%s%s=%d", pos ? " " : "", s->name, static_cast<int32_t>(static_cast<uint32_t>(st)));
    else snprintf(item, sizeof(item), "%s%s=", pos ? " " : "", s->name);
    const int64_t len = static_cast<int64_t>(strlen(item));
    if (pos + len < capacity) {
      memcpy(buf + pos, item, static_cast<size_t>(len) + 1);
      pos += len;
    }
  }
  return n;
}
