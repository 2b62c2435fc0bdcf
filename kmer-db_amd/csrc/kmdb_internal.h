// kmdb_internal.h — shared by the translation units of libkmdb_amd.so (not installed).
#pragma once
#include <string>

// records the message for kmdb_last_error() and returns a non-zero status
int kmdb_set_error(const std::string& msg);
