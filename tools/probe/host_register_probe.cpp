// Diagnostic: which host memory can hipHostRegister page-lock on this box?
// build: hipcc -O2 -o tools/probe/host_register_probe tools/probe/host_register_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

static void
attempt(const char *what, void *p, size_t n, unsigned flags) {
    hipError_t e = hipHostRegister(p, n, flags);
    printf("%-28s flags %#x: %s\n", what, flags, hipGetErrorString(e));
    if (e == hipSuccess) {
	void *d = nullptr;
	hipError_t g = hipHostGetDevicePointer(&d, p, 0);
	printf("    device pointer: %s %p\n", hipGetErrorString(g), d);
	(void) hipHostUnregister(p);
    }
    (void) hipGetLastError();
}

int
main() {
    const size_t n = 64 << 20;
    (void) hipSetDevice(0);
    void *a = nullptr;
    if (posix_memalign(&a, 2 << 20, n) != 0)
	return 1;
    memset(a, 1, n);
    void *b = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    memset(b, 1, n);
    int fd = open("/dev/shm/sora_reg_probe", O_RDWR | O_CREAT, 0600);
    if (ftruncate(fd, n) != 0)
	return 1;
    void *c = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    unlink("/dev/shm/sora_reg_probe");
    memset(c, 1, n);
    void *d = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset(d, 1, n);
    const unsigned flags[] = {hipHostRegisterDefault, hipHostRegisterPortable,
	    hipHostRegisterMapped, hipHostRegisterPortable | hipHostRegisterMapped};
    for (unsigned f : flags) {
	attempt("posix_memalign", a, n, f);
	attempt("mmap private anonymous", b, n, f);
	attempt("mmap shared /dev/shm file", c, n, f);
	attempt("mmap shared anonymous", d, n, f);
    }
    printf("mlock /dev/shm mapping: %d\n", mlock(c, n));
    return 0;
}
