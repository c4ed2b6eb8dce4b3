#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>
#include <dlfcn.h>
static double now(){struct timespec t; clock_gettime(CLOCK_MONOTONIC,&t); return t.tv_sec*1e3+t.tv_nsec*1e-6;}
volatile uint32_t sink[8*64];
void *w(void *a){ long i=(long)a; uint32_t c=1; for(long k=0;k<30000000;k++) c=c*1664525u+1013904223u+(c>>7); sink[i*64]=c; return 0; }
int main(int argc,char**argv){ if(argc>1){void *h=dlopen("/root/repo/pymodem_amd/libpymodem_amd.so", RTLD_NOW); if(!h){puts(dlerror());return 1;}}
 for(int nt=1;nt<=8;nt*=2){ pthread_t t[8]; double t0=now(); for(long i=0;i<nt;i++) pthread_create(&t[i],0,w,(void*)i); for(int i=0;i<nt;i++) pthread_join(t[i],0); printf("%d threads %.2f ms\n",nt,now()-t0);} }
