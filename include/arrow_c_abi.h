/* Arrow C Data Interface / C Device Data Interface struct definitions.
 *
 * These are the ABI-stable definitions the Arrow specification asks producers
 * to embed (https://arrow.apache.org/docs/format/CDataInterface.html,
 * .../CDeviceDataInterface.html); the guard macros make this header
 * interchangeable with <arrow/c/abi.h>.  No libarrow dependency.
 */
#ifndef RUHVRO_ARROW_C_ABI_H
#define RUHVRO_ARROW_C_ABI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE

#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4

struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};

struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif /* ARROW_C_DATA_INTERFACE */

#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE

typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11

struct ArrowDeviceArray {
  struct ArrowArray array;   /* buffers[] hold device pointers */
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event;          /* hipEvent_t* or NULL */
  int64_t reserved[3];
};
#endif /* ARROW_C_DEVICE_DATA_INTERFACE */

#ifdef __cplusplus
}
#endif
#endif
