import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mjlab_amd import robots
from mjlab_amd.sim import Simulation, SimulationCfg
from oracle.oracle import OracleSim
model = robots.load_model("g1_velocity_flat")
nworld = 8
sim = Simulation(nworld, SimulationCfg(njmax=300, use_graph=False), model, "cuda:0")
ora = OracleSim(model, nworld, njmax=300, precision="f64")
rng = np.random.default_rng(0)
qpos = np.tile(model.key_qpos[0], (nworld, 1))
qpos[:, 7:] += rng.normal(0, 0.05, size=(nworld, model.nq - 7))
qpos[:, 2] -= 0.02
qvel = rng.normal(0, 0.1, size=(nworld, model.nv))
ctrl = qpos[:, 7:] + rng.normal(0, 0.1, size=(nworld, model.nu))
sim.data.qpos[:] = torch.from_numpy(qpos.astype(np.float32)).cuda()
sim.data.qvel[:] = torch.from_numpy(qvel.astype(np.float32)).cuda()
sim.data.ctrl[:] = torch.from_numpy(ctrl.astype(np.float32)).cuda()
ora.qpos[:], ora.qvel[:], ora.ctrl[:] = qpos, qvel, ctrl
def rel(a, b):
  a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
  return np.abs(a - b).max() / max(1e-6, np.abs(b).max())
nv = model.nv
for step in range(3):
  sim.forward(); ora.forward(); torch.cuda.synchronize()
  print("step", step, "ncon", sim.data.ncon.cpu().numpy().ravel(), ora.ncon.ravel(), "nefc", sim.data.nefc.cpu().numpy().ravel(), ora.nefc.ravel())
  for w in range(nworld):
    n = int(ora.nefc[w, 0])
    Jg = sim.data.efc_J.cpu().numpy()[w].reshape(-1, nv)[:n]
    Jo = ora.efc_J[w].reshape(-1, nv)[:n]
    errs = {"J": rel(Jg, Jo)}
    for f in ("efc_D", "efc_aref", "efc_pos", "efc_margin"):
      errs[f] = rel(getattr(sim.data, f).cpu().numpy()[w, :n], getattr(ora, f)[w, :n])
    errs["adr"] = int((sim.data.contact_efc_address.cpu().numpy()[w, : int(ora.ncon[w, 0])] != ora.contact_efc_address[w, : int(ora.ncon[w, 0])]).sum())
    errs["qacc"] = rel(sim.data.qacc.cpu().numpy()[w], ora.qacc[w])
    errs["niter"] = (int(sim.data.solver_niter[w]), int(ora.solver_niter[w, 0]))
    print("  w", w, {k: (f"{v:.1e}" if isinstance(v, float) else v) for k, v in errs.items()})
  print("  qpos", rel(sim.data.qpos.cpu().numpy(), ora.qpos), "qvel", rel(sim.data.qvel.cpu().numpy(), ora.qvel))
  sim.step(); ora.step()
