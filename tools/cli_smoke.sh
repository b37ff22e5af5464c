set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cli
python - <<'PY'
import json
doc=json.load(open("goblin_amd/scenes/textured.json"))
doc["camera"]["film"].update({"resolution":[128,128],"file":"gpurun_out/cli/textured.exr","bloom_radius":0.05,"bloom_weight":0.2})
doc["render_setting"]["sample_per_pixel"]=16
for g in doc["geometries"]:
    if "file" in g: g["file"]="../../goblin_amd/scenes/"+g["file"]
json.dump(doc,open("gpurun_out/cli/t.json","w"))
PY
./goblin_amd/lib/g_ray_hip gpurun_out/cli/t.json
ls -la gpurun_out/cli
./goblin_amd/lib/g_ray_hip gpurun_out/cli/t.json --sampler stream --out gpurun_out/cli/textured_stream.ppm
ls -la gpurun_out/cli
