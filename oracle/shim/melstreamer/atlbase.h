// oracle/shim/melstreamer/atlbase.h -- TEST INFRASTRUCTURE ONLY: the ATL names MelStreamer.h uses live in this directory's stdafx.h
#pragma once
#include "stdafx.h"
