/* A C host binding the C ABI of include/ctts.h (no torch, no Python): built and run by tests/test_round6_host_cpu.py on the CPU box.
 * It only calls entry points that need no device: the version, the argument checks of the RCCL-backed collective (SURVEY.md 8(b):
 * ctts_allreduce_* / create / destroy) and the thread-local error text. */
#include <stdio.h>
#include <string.h>
#include "ctts.h"

int main(void) {
  unsigned char id[CTTS_COMM_ID_BYTES];
  void* comm = NULL;
  memset(id, 0, sizeof id);
  if (ctts_version() <= 0) { printf("FAIL version\n"); return 1; }
  if (ctts_comm_create(&comm, 2, 7, id) == 0) { printf("FAIL rank >= nranks accepted\n"); return 1; }
  if (strstr(ctts_last_error(), "ctts_comm_create") == NULL) { printf("FAIL error text: %s\n", ctts_last_error()); return 1; }
  if (ctts_comm_create(NULL, 1, 0, id) == 0) { printf("FAIL null comm_out accepted\n"); return 1; }
  if (ctts_allreduce_mean(NULL, 0, NULL, NULL) != 0) { printf("FAIL empty all-reduce\n"); return 1; }
  if (ctts_allreduce_mean(NULL, 8, NULL, NULL) == 0) { printf("FAIL null buffer accepted\n"); return 1; }
  if (ctts_comm_destroy(NULL) != 0) { printf("FAIL destroy(NULL)\n"); return 1; }
  if (ctts_workspace_bytes() < (size_t)(64u << 20)) { printf("FAIL workspace size\n"); return 1; }
  printf("c host ok: ctts_version %d, workspace %zu bytes\n", ctts_version(), ctts_workspace_bytes());
  return 0;
}
