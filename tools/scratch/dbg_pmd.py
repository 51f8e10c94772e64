import os, sys
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
os.environ["HRX_DEBUG_FLAGS"]=str(0x4000000|0x20000000)
import numpy as np, torch
import halo2_regex_amd as hra
from halo2_regex_amd import synth
from oracle_lib import OracleDefs, DFA_DIR, load_oracle
CFG_123=[["regex1_test_lookup.txt",["substr1_test_lookup.txt"]],["regex2_test_lookup.txt",["substr2_test_lookup.txt"]],["regex3_test_lookup.txt",["substr3_test_lookup.txt"]]]
names=CFG_123; M=704
chars, lens = synth.reveal_stress(700, 700, seed=21)
chars[5,100]=200; chars[6,699]=255; chars[7,650]=128; lens[7]=600
defs=[hra.RegexDefs(hra.AllstrRegexDef.read_from_text(os.path.join(DFA_DIR,a)),[hra.SubstrRegexDef.read_from_text(os.path.join(DFA_DIR,s)) for s in subs]) for a,subs in names]
cfg=hra.RegexVerifyConfig.configure(M,defs,device=0)
print(cfg.describe_launch(700,layout=1))
o=OracleDefs.from_files(load_oracle(),names)
orec,omsk,ost=o.witness_batch(chars,lens,M)
dev=torch.device("cuda",0); B=700; D=3
stride=(chars.shape[1]+15)//16*16
wide=torch.zeros((B,max(stride,16)),dtype=torch.uint8,device=dev); wide[:,:chars.shape[1]]=torch.from_numpy(chars).to(dev)
d_lens=torch.from_numpy(lens.astype(np.int32)).to(dev)
for rep in range(2):
    rec,msk,st=cfg.witness_batch_position_major(wide,d_lens)
    rec_i,msk_i,st_i=cfg.witness_batch_position_major(hra.chars_to_position_major(wide),d_lens,chars_pm_stride=wide.shape[1])
    torch.cuda.synchronize()
    a_r,a_m=hra.position_major_to_string_major(rec,msk,B,M,D); b_r,b_m=hra.position_major_to_string_major(rec_i,msk_i,B,M,D)
    ok=(ost&np.uint64(0xff))==0
    for nm,m in (("sm-input",a_m),("pm-input",b_m)):
        g=m.cpu().numpy().view(np.uint16)
        bad=np.argwhere((g!=omsk)&ok[:,None])
        print(rep,nm,"mismatching cells",len(bad))
        for b,r in bad[:12]: print("   string",b,"row",r,"len",lens[b],"got %04x want %04x"%(g[b,r],omsk[b,r]),"char",chars[b,r] if r<chars.shape[1] else None)
