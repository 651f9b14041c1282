#!/usr/bin/env python3
"""Which HIP / RCCL / HSA copies end up in one process, and does torch still see the GPU, for the load orders the tests and bench.py use."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "rxgpu_comm_then_torch": "from reindexer_amd import capi; import numpy as np\nsx=capi.ShardedVectorIndex(1,16,256,[0,0]); sx.upload_rows(0,np.ones((256,16),np.float32)); print('mode',sx.merge_mode, sx.search_knn(np.ones((1,16),np.float32),3)[1]); sx.close()\nimport torch; torch.cuda.init(); print('torch sees', torch.cuda.device_count())",
    "rxgpu_then_torch": "from reindexer_amd import capi\ncapi.lib(); print(capi.device_count())\nimport torch; torch.cuda.init(); print('torch sees', torch.cuda.device_count())",
    "torch_then_rxgpu_comm": "import torch; torch.cuda.init(); print('torch sees', torch.cuda.device_count())\nfrom reindexer_amd import capi; import numpy as np\nsx=capi.ShardedVectorIndex(1,16,256,[0,0]); sx.upload_rows(0,np.ones((256,16),np.float32)); print('mode',sx.merge_mode, sx.search_knn(np.ones((1,16),np.float32),3)[1]); sx.close()",
    "rxgpu_comm_open_then_torch": "from reindexer_amd import capi; import numpy as np\nsx=capi.ShardedVectorIndex(1,16,256,[0,0]); sx.upload_rows(0,np.ones((256,16),np.float32)); print('mode',sx.merge_mode)\nimport torch; torch.cuda.init(); print('torch sees', torch.cuda.device_count()); print(sx.search_knn(np.ones((1,16),np.float32),3)[1]); sx.close()",
}
TAIL = "\nimport re\nlibs=sorted({l.split()[-1] for l in open('/proc/self/maps') if re.search(r'amdhip64|rccl|hsa-runtime', l)})\nprint('LIBS', libs)\nimport os\nprint('ENV', {k:v for k,v in os.environ.items() if 'VISIBLE' in k or k.startswith('HSA_') or k.startswith('NCCL') or k.startswith('RCCL')})"
for name, code in CASES.items():
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code + TAIL], capture_output=True, text=True, timeout=300)
    print("=====", name, "rc", r.returncode)
    print(r.stdout[-1500:])
    print(r.stderr[-800:])
