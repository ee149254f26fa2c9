// dev-only: per-node per-ring degree histograms on CPU (BFS), OpenMP
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>
int main(int argc,char**argv){
  int64_t n,nnz; int K=atoi(argv[2]);
  FILE*f=fopen(argv[1],"rb"); fread(&n,8,1,f); fread(&nnz,8,1,f);
  int64_t*rp=malloc((n+1)*8); int32_t*col=malloc(nnz*4); fread(rp,8,n+1,f); fread(col,4,nnz,f); fclose(f);
  int32_t*deg=malloc(n*4); int maxd=0; for(int64_t i=0;i<n;i++){deg[i]=rp[i+1]-rp[i]; if(deg[i]>maxd)maxd=deg[i];}
  int*bin=calloc(maxd+1,4); for(int64_t i=0;i<n;i++)bin[deg[i]]=1; int B=0; for(int d=0;d<=maxd;d++){ if(bin[d]){bin[d]=B++;} else bin[d]=-1;}
  fprintf(stderr,"n %ld nnz %ld B %d maxd %d\n",n,nnz,B,maxd);
  uint16_t*H=calloc((size_t)n*K*B,2); // rings 1..K
  #pragma omp parallel
  { int32_t*q=malloc(n*4); int32_t*mark=malloc(n*4); memset(mark,0xff,n*4);
    #pragma omp for schedule(dynamic,64)
    for(int64_t s=0;s<n;s++){ int qh=0,qt=0; q[qt++]=s; mark[s]=s; 
      for(int k=1;k<=K;k++){ int lev_end=qt; uint16_t*h=H+((size_t)s*K+(k-1))*B;
        for(;qh<lev_end;qh++){int v=q[qh]; for(int64_t e=rp[v];e<rp[v+1];e++){int u=col[e]; if(mark[u]!=s){mark[u]=s; q[qt++]=u; h[bin[deg[u]]]++;}}}
      }
    }
  }
  f=fopen(argv[3],"wb"); fwrite(&B,4,1,f); int*sup=malloc(B*4); for(int d=0;d<=maxd;d++) if(bin[d]>=0) sup[bin[d]]=d; fwrite(sup,4,B,f); fwrite(H,2,(size_t)n*K*B,f); fclose(f);
  return 0;}
