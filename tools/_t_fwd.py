import sys, os
sys.path.insert(0, "/root/repo")
import torch, kokoro_align_amd as ka
from kokoro_align_amd.align import DeviceBatch
B,T,V,S=8192,50000,64,5000
lib=ka.load_library()
lps=torch.empty((B,T,V),dtype=torch.float32,device="cuda"); labs=torch.empty((B,S),dtype=torch.int32,device="cuda")
lib.ka_hash_logprobs_batch_f32(lps.data_ptr(),B,T,V,V,T*V,0,None); lib.ka_hash_labels_batch_i32(labs.data_ptr(),B,S,V,S,0,None); torch.cuda.synchronize()
b=DeviceBatch([lps[i] for i in range(B)],[labs[i] for i in range(B)]); b.engine.set_profiling(True)
for _ in range(2): b.run(raise_on_error=False)
print(os.path.basename(os.environ.get("KA_LIBRARY","default")), b.engine.last_kernel_ms())
