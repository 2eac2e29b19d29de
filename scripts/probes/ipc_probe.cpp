// What two processes on ONE GPU can share (the peer exchange backend): device memory through
// hipIpcMemHandle, ordering through interprocess events and through stream memory operations on shared device memory.
//   hipcc --offload-arch=gfx950 -o ipc_probe ipc_probe.cpp && ./ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%s] %s failed: %s\n", who, #x, hipGetErrorString(e_)); fflush(stdout); _exit(2); } } while (0)
struct Msg { hipIpcMemHandle_t mem, flag; hipIpcEventHandle_t ev; };
__global__ void k_fill(unsigned *p, int n, unsigned v) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v + i; }
int main(int argc, char **argv)
{
   setvbuf(stdout, nullptr, _IONBF, 0);
   const int mode = argc > 1 ? atoi(argv[1]) : 0; // 0: host-ordered only; 1: + the device-side wait enqueued AFTER the write; 2: enqueued BEFORE it
   int a2b[2], b2a[2];
   if (pipe(a2b) || pipe(b2a)) return 1;
   const int N = 1 << 20;
   pid_t pid = fork(); // (before anything touches the GPU)
   if (pid == 0) { // B: opens A's memory, fills it, signals
      const char *who = "B";
      Msg m;
      if (read(a2b[0], &m, sizeof m) != (ssize_t)sizeof m) _exit(3);
      CK(hipSetDevice(0));
      unsigned *peer = nullptr, *flag = nullptr;
      CK(hipIpcOpenMemHandle((void **)&peer, m.mem, hipIpcMemLazyEnablePeerAccess));
      CK(hipIpcOpenMemHandle((void **)&flag, m.flag, hipIpcMemLazyEnablePeerAccess));
      hipEvent_t ev;
      hipError_t ee = hipIpcOpenEventHandle(&ev, m.ev);
      printf("[B] hipIpcOpenEventHandle: %s\n", hipGetErrorString(ee));
      hipStream_t st;
      CK(hipStreamCreate(&st));
      unsigned *mine = nullptr;
      CK(hipMalloc(&mine, N * 4));
      hipLaunchKernelGGL(k_fill, dim3(N / 256), dim3(256), 0, st, mine, N, 1000u);
      CK(hipMemcpyAsync(peer, mine, N * 4, hipMemcpyDeviceToDevice, st)); // my slice into the peer's buffer
      printf("[B] opened A's memory and copied into it\n");
      if (mode >= 1) {
         hipError_t we = hipStreamWriteValue32(st, flag, 7u, 0);
         printf("[B] hipStreamWriteValue32 on the peer's memory: %s\n", hipGetErrorString(we));
      }
      if (ee == hipSuccess) { hipError_t re = hipEventRecord(ev, st); printf("[B] hipEventRecord on the opened event: %s\n", hipGetErrorString(re)); }
      CK(hipStreamSynchronize(st));
      char ok = 1;
      if (write(b2a[1], &ok, 1) != 1) _exit(3);
      char done;
      if (read(a2b[0], &done, 1) != 1) _exit(3);
      CK(hipIpcCloseMemHandle(peer));
      CK(hipIpcCloseMemHandle(flag));
      _exit(0);
   }
   const char *who = "A";
   CK(hipSetDevice(0));
   unsigned *buf = nullptr, *flag = nullptr;
   CK(hipMalloc(&buf, N * 4));
   CK(hipMalloc(&flag, 4096));
   CK(hipMemset(buf, 0, N * 4));
   CK(hipMemset(flag, 0, 4096));
   hipEvent_t ev;
   CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
   Msg m;
   CK(hipIpcGetMemHandle(&m.mem, buf));
   CK(hipIpcGetMemHandle(&m.flag, flag));
   hipError_t ge = hipIpcGetEventHandle(&m.ev, ev);
   printf("[A] hipIpcGetEventHandle: %s\n", hipGetErrorString(ge));
   hipStream_t st;
   CK(hipStreamCreate(&st));
   unsigned *host = (unsigned *)malloc(N * 4);
   if (mode == 2) { // A waits ON THE DEVICE for B's flag before it reads the buffer: enqueued BEFORE B has done anything
      hipError_t wv = hipStreamWaitValue32(st, flag, 7u, hipStreamWaitValueGte, 0xffffffffu);
      printf("[A] hipStreamWaitValue32 enqueued early: %s\n", hipGetErrorString(wv));
      CK(hipMemcpyAsync(host, buf, N * 4, hipMemcpyDeviceToHost, st));
   }
   if (write(a2b[1], &m, sizeof m) != (ssize_t)sizeof m) return 3;
   if (mode == 2) {
      hipError_t se = hipStreamSynchronize(st);
      printf("[A] stream with the early device-side wait finished: %s; buf[5] = %u (want 1005), buf[N-1] = %u (want %u)\n", hipGetErrorString(se), host[5], host[N - 1], 1000u + N - 1);
   }
   char ok;
   if (read(b2a[0], &ok, 1) != 1) return 3;
   unsigned f = 0;
   CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
   CK(hipMemcpy(host, buf, N * 4, hipMemcpyDeviceToHost));
   printf("[A] after B's host said done: flag = %u, buf[5] = %u (want 1005)\n", f, host[5]);
   if (mode == 1) {
      hipError_t wv = hipStreamWaitValue32(st, flag, 7u, hipStreamWaitValueGte, 0xffffffffu);
      printf("[A] hipStreamWaitValue32 enqueued late: %s\n", hipGetErrorString(wv));
      hipError_t se = hipStreamSynchronize(st);
      printf("[A] stream with the late device-side wait finished: %s\n", hipGetErrorString(se));
   }
   if (ge == hipSuccess) { hipError_t q = hipEventQuery(ev); printf("[A] hipEventQuery of the interprocess event: %s\n", hipGetErrorString(q)); }
   char done = 1;
   if (write(a2b[1], &done, 1) != 1) return 3;
   int status = 0;
   waitpid(pid, &status, 0);
   printf("[A] B exited with %d\n", WEXITSTATUS(status));
   return 0;
}
