// uid_exchange_main.cpp — test harness (tests/test_uid_exchange.py): runs the drivers' RCCL
// unique-id file handshake (host/driver_common.h: exchange_uid) with a stand-in id generator, so
// the rendezvous logic is exercised on a CPU-only box.  usage: uid_exchange_main <rank> <world>
#include "../pairwise-perturbation_amd/host/driver_common.h"

extern "C" int ppals_get_unique_id(void *out128) {
  unsigned char *p = (unsigned char *)out128;
  const uint64_t pid = (uint64_t)getpid();
  const uint64_t t = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                         std::chrono::system_clock::now().time_since_epoch()).count();
  for (int i = 0; i < PPALS_UNIQUE_ID_BYTES; i++)
    p[i] = (unsigned char)((i < 8 ? pid >> (8 * i) : t >> (8 * (i % 8))) + i);
  return 0;
}
extern "C" const char *ppals_last_error(void) { return ""; }

int main(int argc, char **argv) {
  const int rank = atoi(argv[1]), world = atoi(argv[2]);
  if (argc > 3) usleep(1000 * atoi(argv[3]));  // start-up skew in ms
  unsigned char uid[PPALS_UNIQUE_ID_BYTES];
  if (exchange_uid(rank, world, uid) != 0) {
    printf("FAIL\n");
    return 1;
  }
  for (int i = 0; i < PPALS_UNIQUE_ID_BYTES; i++) printf("%02x", uid[i]);
  printf("\n");
  fflush(stdout);
  usleep(50000);  // stands for ncclCommInitRank (collective)
  uid_exchange_done(rank);
  return 0;
}
