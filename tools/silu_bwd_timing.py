import os, sys, torch
sys.path.insert(0, "/root/repo")
from sde_sampler_amd import problems
for name, method in (("cfg3_gmm50_pis_kl","kl"),("cfg1_dw_dis_lv","lv")):
    spec = problems.baseline_spec(name); spec["batch"]=65536; spec["net"]["activation"]="silu"; spec["loss"]["method"]=method
    prob = problems.build(spec, device="cuda:0"); eng=prob.loss.engine; eng.timing=True
    x0 = prob.prior.sample((65536,)); tb=[]
    for rep in range(6):
        prob.ctrl.zero_grad(); val,_=prob.loss(prob.ts,x0,prob.target.unnorm_log_prob,prob.second_log_prob); torch.cuda.synchronize()
        val.backward(); torch.cuda.synchronize(); tb.append(eng.last_kernel_ms())
    print(name, method, "silu backward", eng.last_kernel_name(), sorted(tb)[3])
