// part of the host emulation (see hip_runtime.h in this directory)
#pragma once
