#include "../../hip_emu.hpp"
