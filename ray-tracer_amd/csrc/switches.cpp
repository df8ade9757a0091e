// switches.cpp -- see switches.hpp
#include "switches.hpp"

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace mi355rt {

namespace {

std::mutex g_mutex;
bool g_parsed = false;
Switches g_env;   // what the environment said at first use
Switches g_cur;

int parse_transport(const char *v) { return !v || !*v ? 0 : (v[0] == 'p' ? 1 : (v[0] == 'r' ? 2 : 0)); }

// applies one (name, value) pair to s; value == nullptr restores the built-in default
bool apply(Switches &s, const char *name, const char *value)
{
   static const Switches def;
   if (std::strncmp(name, "MI355RT_", 8) == 0) name += 8;
   const bool has = value && *value;
   auto num = [&](int fallback) { return has ? std::atoi(value) : fallback; };
   if (!std::strcmp(name, "BUILDER")) s.builder_host = has && !std::strcmp(value, "host") ? 1 : 0;
   else if (!std::strcmp(name, "BUILD_TIMING")) s.build_timing = has ? 1 : 0;
   else if (!std::strcmp(name, "FUSE_SHADOW")) s.fuse_shadow = has ? (value[0] == '0' ? 0 : (value[0] == '1' ? 1 : -1)) : def.fuse_shadow;
   else if (!std::strcmp(name, "PASS_SLOTS_LOG2")) s.pass_slots_log2 = num(def.pass_slots_log2);
   else if (!std::strcmp(name, "BIG_LEAVES")) s.big_leaves = has ? (std::atoi(value) != 0 ? 1 : 0) : def.big_leaves;
   else if (!std::strcmp(name, "BLOCKS_PER_CU")) s.blocks_per_cu = num(def.blocks_per_cu);
   else if (!std::strcmp(name, "BLOCKS_PER_CU_SHADOW")) s.blocks_per_cu_shadow = num(def.blocks_per_cu_shadow);
   else if (!std::strcmp(name, "TRIM")) s.trim = num(def.trim);
   else if (!std::strcmp(name, "TRIM_PRIMARY_X2")) s.trim_primary_x2 = num(def.trim_primary_x2);
   else if (!std::strcmp(name, "TRIM_SHADOW_X2")) s.trim_shadow_x2 = num(def.trim_shadow_x2);
   else if (!std::strcmp(name, "SPREAD_SL")) s.spread_sl = num(def.spread_sl);
   else if (!std::strcmp(name, "SPREAD_BL")) s.spread_bl = num(def.spread_bl);
   else if (!std::strcmp(name, "SPREAD_PRIMARY_MAX")) s.spread_primary_max = num(def.spread_primary_max);
   else if (!std::strcmp(name, "TRANSPORT")) s.transport = parse_transport(value);
   else if (!std::strcmp(name, "BOUNCE_SYNC")) s.bounce_sync = num(def.bounce_sync);
   else if (!std::strcmp(name, "GPU_GROUP")) s.device_group = has ? (std::atoi(value) != 0 ? 1 : 0) : def.device_group;
   else return false;
   return true;
}

void parse_locked()
{
   if (g_parsed) return;
   static const char *names[] = { "BUILDER", "BUILD_TIMING", "FUSE_SHADOW", "PASS_SLOTS_LOG2", "BIG_LEAVES", "BLOCKS_PER_CU", "BLOCKS_PER_CU_SHADOW",
                                  "TRIM", "TRIM_PRIMARY_X2", "TRIM_SHADOW_X2", "SPREAD_SL", "SPREAD_BL", "SPREAD_PRIMARY_MAX", "TRANSPORT", "BOUNCE_SYNC",
                                  "GPU_GROUP" };
   for (const char *n : names) {
      char full[64] = "MI355RT_";
      std::strncat(full, n, sizeof(full) - 9);
      const char *v = std::getenv(full);
      if (v && *v) (void) apply(g_env, n, v);
   }
   g_cur = g_env;
   g_parsed = true;
}

}   // namespace

Switches switches()
{
   std::lock_guard<std::mutex> lock(g_mutex);
   parse_locked();
   return g_cur;
}

bool set_switch(const char *name, const char *value)
{
   std::lock_guard<std::mutex> lock(g_mutex);
   parse_locked();
   if (!name) {
      g_cur = g_env;
      return true;
   }
   return apply(g_cur, name, value);
}

}   // namespace mi355rt
