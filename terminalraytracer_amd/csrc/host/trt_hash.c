/* trt_hash.c -- FNV-1a-64 of a byte range: the frame fingerprint the goldens are recorded with (offset
 * 1469598103934665603, prime 1099511628211, over the raw little-endian bytes of pixels[0..W*H)).  Host utility of the
 * product library so that a driver (bench.py, examples/trt_demo.c) can check the frame it just produced. */
#include "trt_host.h"

unsigned long long trt_fnv1a64(const void *data, size_t bytes)
{
    const unsigned char *p = (const unsigned char *)data;
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < bytes; i++)
        h = (h ^ p[i]) * 1099511628211ull;
    return h;
}
