// oracle/shim/melstreamer/mfreadwrite.h -- TEST INFRASTRUCTURE ONLY: Media Foundation is replaced by the memory reader of mfidl.h; the other names live in this directory's stdafx.h
#pragma once
#include "stdafx.h"
