// oracle/shim/melstreamer/mfapi.h -- TEST INFRASTRUCTURE ONLY: MFllMulDiv lives in this directory's stdafx.h
#pragma once
#include "stdafx.h"
