// kconfig.hpp -- launch constants shared by host launchers and device code (also compiled by hiprtc: no host headers).
#pragma once
namespace plx {
namespace k {
constexpr int kBlock = 256;  // 4 wave64 per workgroup
}  // namespace k
}  // namespace plx
